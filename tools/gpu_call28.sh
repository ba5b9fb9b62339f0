cd $GRAFT_REPO_ROOT
python tools/dbg_deform_patch.py 2>&1 | grep -v amdgpu.ids | grep "bad frac\|couts"
timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -10
