set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5final_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5final_pytest.log
grep -v "^  File\|^$" gpurun_out/r5final_pytest.log | tail -n 5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/profile_round5.sh 2>&1 | tail -n 12
