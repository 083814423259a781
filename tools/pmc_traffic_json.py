"""gpurun_out/pmcb[_x6]_{FETCH_SIZE,WRITE_SIZE}/p_counter_collection.csv -> the json bench.py reads
(profiles/<round>_pmc_gemm_traffic.json, <round>_pmc_x6_traffic.json): HBM bytes per launch of the dominant kernel
family -- MODE=fp32 (default): the exact-fp32 lean GEMM; MODE=bf16x6: the six-product GEMM kernels."""
import csv, json, os, re, sys
sys.path.insert(0, os.getcwd())
G = os.environ.get("G", "gpurun_out")
MODE = os.environ.get("MODE", "fp32")
if MODE == "bf16x6":
    pat, sub, gemm, family = r"gemm_(x6p|x6g|x6f|x6|leanw6t|leanw6s|leanw6)_kernel", "pmcb_x6", " --gemm bf16x6", "x6"
    label = "gemm_x6p / x6 / x6g / x6f / leanw6 / leanw6t / leanw6s kernels (fp32 class on the bf16 pipe, the `x6` family of bench.py)"
else:
    pat, sub, gemm, family = r"gemm_lean_kernel<(false|0), \d, 0[,>]", "pmcb", " --gemm fp32", "lean"
    label = "gemm_lean_kernel<0, EP, 0> (exact fp32, all epilogue instances)"
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(f"{G}/{sub}_{name}/p_counter_collection.csv")):
        if re.search(pat, row["Kernel_Name"]):
            tot += float(row["Counter_Value"])
            n += 1
    res[name] = (tot, n)
f, nf = res["FETCH_SIZE"]
w, nw = res["WRITE_SIZE"]
from flow2gan_amd import _lib
json.dump({"source": "F2G_STREAMS=0 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py"
                     + gemm + " --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode; dispatches of the kernel(s) below (2 steps)",
           "lib_version": _lib.version(), "kernel": label, "family": family,
           "launches": nf, "fetch_kib_per_launch": f / nf, "write_kib_per_launch": w / nw,
           "hbm_bytes_per_launch_raw": (f / nf + w / nw) * 1024, "hbm_bytes_per_launch_fetch_x2": (2 * f / nf + w / nw) * 1024,
           "note": "gfx950 FETCH_SIZE under-reports wide coalesced reads by up to 2x (MI355X_MICROARCH.md, HBM); both raw and "
                   "x2-corrected sums given (rocprofv3 reports these counters in KiB)"},
          open(sys.argv[1], "w"), indent=1)
