"""Kernel timeline of the LAST inference in a rocprofv3 --kernel-trace csv (start us, duration us, queue,
kernel): python tools/timeline_last.py <p_kernel_trace.csv> [first_kernel_prefix] [max_rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "bct_to_rows"
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ev = []
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)[:40]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), grid // max(wg, 1)))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith(first)]
seg = ev[idx[-1]:] if idx else ev
t0 = seg[0][0]
print(f"last inference: {len(seg)} kernels, wall {(max(e[1] for e in seg) - t0) / 1e3:.1f} us")
for e in seg[:nmax]:
    print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:7.1f} q{e[3]} {e[4]:6d} {e[2]}")
