# FeatureAlign kernel choice at step level on the SSD-style shape: LDS-window kernel vs gather loader
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pass in 1 2; do
  for g in 0 1; do
    SIPMASK_DEFORM_GATHER=$g timeout 300 python bench.py --config ssd --steps 400 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c17_g${g}_$pass.json 2> gpurun_out/r5c17_g${g}_$pass.err
    echo "gather=$g pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c17_g${g}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['deform_kernel'])")"
  done
done
