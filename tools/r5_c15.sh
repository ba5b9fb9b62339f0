# bench.py --config ssd (544 x 544, 2-conv towers without GN, fast_nms, 8 images per GPU): first run
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --config ssd --cpu-budget 8 --breakdown gpurun_out/r5c15_ssd_breakdown.txt > gpurun_out/r5c15_ssd.json 2> gpurun_out/r5c15_ssd.err; echo "ssd rc $?"
tail -n 5 gpurun_out/r5c15_ssd.err | cut -c1-300
cut -c1-1500 gpurun_out/r5c15_ssd.json
grep -v '^conv:backbone' gpurun_out/r5c15_ssd_breakdown.txt | head -50
timeout 300 python bench.py --config ssd --det-boxes tiny --no-cpu-baseline --no-extras --batch 4 | cut -c1-400
