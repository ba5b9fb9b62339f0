cd $GRAFT_REPO_ROOT
for A in "" "--sub-graphs 2" "--sub-graphs 2 --free-run 1" "--sub-graphs 2 --free-run 2" "--sub-graphs 2 --free-run 2 --steps 100"; do
  echo "== $A"
  timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 $A 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['steps'], d['config']['detections_per_image'])
except Exception as e:
    print('ERR', l[-600:])"
done
