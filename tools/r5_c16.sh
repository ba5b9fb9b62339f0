# bench.py --config ssd with the offset calibration; wall clock of the complete default line (other_configs now has four children)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --config ssd --cpu-budget 8 --breakdown gpurun_out/r5c16_ssd_breakdown.txt > gpurun_out/r5c16_ssd.json 2> gpurun_out/r5c16_ssd.err; echo "ssd rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c16_ssd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steady_state']['value'], d['with_results']['value'], d['config']['deform_kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])
PY
grep 'feat_align\|tower\|cls_convs\|reg_convs' gpurun_out/r5c16_ssd_breakdown.txt
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5c16_default.json 2> gpurun_out/r5c16_default.err; echo "default rc $? wall ${SECONDS}s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c16_default.json').read().strip().splitlines()[-1])
print(d['value'], d['steady_state']['value'], d['with_results']['value'], [ (p['plan'], p['img_s'], p['mask_logit_max_abs_features']) for p in d['parity_pairs']])
print({k:(v.get('value'), v.get('ms_per_step'), v.get('skipped'), v.get('error')) for k,v in d['other_configs'].items()})
PY
