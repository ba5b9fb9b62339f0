#!/usr/bin/env python
"""aggregate a rocprofv3 --kernel-trace results db into a per-kernel csv (calls, total/avg/min/max us, pct)"""
import collections, glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
agg = collections.defaultdict(list)
for kid, s, e in cur.execute("select kernel_id, start, end from %s" % kd):
    agg[names[kid]].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
out = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    short = re.sub(r"\(anonymous namespace\)::", "", k)[:120]
    out.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (short, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]))
