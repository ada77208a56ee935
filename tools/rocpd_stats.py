"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into a per-kernel stats table (markdown-ish text).
usage: python tools/rocpd_stats.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db.split('/')[-1]} (durations in ms; total GPU kernel time {tot/1e6:.1f} ms)",
             f"{'kernel':112s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}"]
    for n, k, s, a, mn, mx in rows:
        lines.append(f"{short(n):112s} {k:6d} {s/1e6:10.3f} {a/1e6:9.4f} {mn/1e6:9.4f} {mx/1e6:9.4f} {100*s/tot:6.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
