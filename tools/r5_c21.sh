set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r5c21_marginal.txt
bash tools/marginal_cost.sh gpurun_out/r5c21_marginal.txt 1
