"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel calls / total / avg / share."""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kcols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in kcols else "kernel_name"
rows = c.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                 f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, cnt, s, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n if len(n) <= 70 else n[:67] + "..."
    print(f"{n:70s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
