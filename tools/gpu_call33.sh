cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_deform_patch.py -x -q 2>&1 | tail -2
timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -10
for A in 2 4; do
  echo "== ablate $A"; SIPMASK_DEFORM_ABLATE=$A timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep "N(0,0.3)" | cut -c1-62
done
