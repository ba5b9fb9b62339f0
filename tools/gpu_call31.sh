cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_deform_patch.py -x -q 2>&1 | tail -2
echo "== packed blend"; timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -10
echo "== scalar blend"; SIPMASK_DEFORM_SCALAR_BLEND=1 timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -10 | cut -c1-60
