"""gpurun_out/pmcb_{FETCH_SIZE,WRITE_SIZE}/p_counter_collection.csv -> the json bench.py reads
(profiles/<round>_pmc_gemm_traffic.json): HBM bytes per launch of the exact-fp32 lean GEMM kernel."""
import csv, json, os, re, sys
sys.path.insert(0, os.getcwd())
G = "gpurun_out"
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(f"{G}/pmcb_{name}/p_counter_collection.csv")):
        if re.search(r"gemm_lean_kernel<(false|0), \d, 0[,>]", row["Kernel_Name"]):   # exact-fp32 instances
            tot += float(row["Counter_Value"])
            n += 1
    res[name] = (tot, n)
f, nf = res["FETCH_SIZE"]
w, nw = res["WRITE_SIZE"]
from flow2gan_amd import _lib
json.dump({"source": "F2G_STREAMS=0 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                     "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode; gemm_lean_kernel<0, EP, 0> dispatches (2 steps)",
           "lib_version": _lib.version(), "kernel": "gemm_lean_kernel<0, EP, 0> (exact fp32, all epilogue instances)",
           "launches": nf, "fetch_kib_per_launch": f / nf, "write_kib_per_launch": w / nw,
           "hbm_bytes_per_launch_raw": (f / nf + w / nw) * 1024, "hbm_bytes_per_launch_fetch_x2": (2 * f / nf + w / nw) * 1024,
           "note": "gfx950 FETCH_SIZE under-reports wide coalesced reads by up to 2x (MI355X_MICROARCH.md, HBM); both raw and "
                   "x2-corrected sums given (rocprofv3 reports these counters in KiB)"},
          open(sys.argv[1], "w"), indent=1)
