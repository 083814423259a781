"""Per-kernel means from one rocprofv3 --pmc + --kernel-trace pass (tools/pmc_busy.sh): launches, duration,
effective clock = GRBM_GUI_ACTIVE / duration, matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), wait share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for row in csv.DictReader(open(kt)):
    dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for row in csv.DictReader(open(cc)):
    k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Dispatch_Id"] not in seen[k]:
        seen[k].add(row["Dispatch_Id"])
        agg[k]["_dur"] += dur.get(row["Dispatch_Id"], 0.0)
rows = []
for k, c in agg.items():
    n = len(seen[k])
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    if g <= 0 or c["_dur"] <= 0:
        continue
    rows.append((c["_dur"], k, n, c["_dur"] / n * 1e6, g / c["_dur"] / 1e9,
                 c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8 * 1024),
                 c.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1.0)),
                 c.get("SQ_INSTS_VALU", 0.0) / max(1.0, c.get("SQ_INSTS_MFMA", 1.0))))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'n':>5s} {'total_ms':>9s} {'avg_us':>8s} {'GHz':>5s} {'mfma_busy':>9s} {'wait':>5s} {'valu/mfma':>9s}")
for t, k, n, us, ghz, busy, wait, vm in rows[:40]:
    print(f"{k[:58]:58s} {n:5d} {t * 1e3:9.2f} {us:8.1f} {ghz:5.2f} {busy:9.3f} {wait:5.2f} {vm:9.2f}")
