"""Per-kernel PMC sums from a rocprofv3 rocpd db.  usage: rocpd_pmc.py db [kernel-substring]"""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else ("name" if "name" in ix else None)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    kn = re.sub(r"\(anonymous namespace\)::|void |mhattn::|mhgemm::", "", str(r[ix[name_col]]))[:60]
    if flt and flt not in kn: continue
    agg[kn][r[ix["counter_name"]]] += float(r[ix["value"]]); cnt[(kn, r[ix["counter_name"]])] += 1
for kn, d in agg.items():
    print(kn)
    for cn, v in sorted(d.items()):
        print(f"   {cn:34s} {v / max(1, cnt[(kn, cn)]):16.1f}  (avg over {cnt[(kn, cn)]} dispatches)")
