cd $GRAFT_REPO_ROOT
for A in 0 1 2 3 4; do
  echo "== ablate $A"; SIPMASK_DEFORM_ABLATE=$A timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep "N(0,0.3)" | cut -c1-62
done
