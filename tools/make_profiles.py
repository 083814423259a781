"""Turn the raw outputs of tools/collect_profiles.sh (gpurun_out/) into profiles/<round>_*
(ROUND=r02 by default)."""
import csv, json, re, os, sys
G, P = "gpurun_out", "profiles"
R = os.environ.get("ROUND", "r04")
sys.path.insert(0, os.getcwd())


def stats(path, out, header):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    L = [header, f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/3:.1f} ms per step (3 steps in the trace)",
         f"{'kernel':66s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}"]
    for r in rows[:48]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:66]
        L.append(f"{n:66s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} "
                 f"{float(r['MinNs'])/1e3:8.1f} {float(r['MaxNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
    g = [r for r in rows if "gemm_kernel" in r["Name"] or "gemm_lean" in r["Name"] or "narrow_" in r["Name"]
         or "conv32_" in r["Name"] or "gemm_x6" in r["Name"]]
    gc = sum(int(r["Calls"]) for r in g); gt = sum(float(r["TotalDurationNs"]) for r in g)
    L.append(f"# all gemm_lean / gemm_x6* / gemm_kernel<...> / narrow_* / conv32_* dispatches (= the MFMA-class launches bench.py times; the thin first / last discriminator layers -- conv2ch_*, convpost_*, mpd0_*, mpdpost_* -- are HBM class): {gc} calls, {gt/1e6:.1f} ms, average {gt/gc/1e3:.1f} us, "
             f"{100*gt/tot:.1f} % of kernel time")
    open(out, "w").write("\n".join(L) + "\n")
    print(L[1]); print(L[-1])


CMD = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
stats(f"{G}/prof_serial/p_kernel_stats.csv", f"{P}/{R}_gan_stage2_kernel_stats.txt",
      f"# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- {CMD}\n"
      "# (mel_24k_base GAN stage-2 step, B=64 x 1 s per D-step and per G-step, exact-fp32 GEMMs, 1 x MI355X)\n"
      "# launch lanes OFF: every kernel alone on the chip -- the same condition as bench.py's roofline pass, whose per-launch average must agree")
stats(f"{G}/prof_lanes/p_kernel_stats.csv", f"{P}/{R}_gan_stage2_kernel_stats_lanes.txt",
      f"# rocprofv3 --kernel-trace --stats -- {CMD}\n"
      "# default mode: launch lanes ON (up to 7 HIP streams): kernel durations overlap and stretch, their sum exceeds the wall time of a step")
stats(f"{G}/prof_b3/p_kernel_stats.csv", f"{P}/{R}_fast_mode_kernel_stats.txt",
      f"# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- {CMD} --gemm bf16x3\n"
      "# the same step in the split-bf16 fast mode (pre-split operand images; lean / K-major weight-gradient / direct-conv kernels), launch lanes OFF")
if os.path.exists(f"{G}/prof_x6/p_kernel_stats.csv"):
    stats(f"{G}/prof_x6/p_kernel_stats.csv", f"{P}/{R}_x6_mode_kernel_stats.txt",
          f"# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- {CMD} --gemm bf16x6\n"
          "# the same step with fp32-class products on the bf16 pipe for the long-reduction GEMMs (gemm_x6_kernel: forward / data gradient over three-piece images; "
          "gemm_leanw6_kernel: weight gradients, operands split in the kernel), launch lanes OFF")
if os.path.exists(f"{G}/x6_shapes.txt"):
    def table(path, n):
        t = [l for l in open(path).read().split("\n") if "amdgpu.ids" not in l]
        i = next(k for k, l in enumerate(t) if l.startswith("form"))
        return "\n".join(t[i:i + n + 1])
    fam = json.loads(open(f"{G}/x6_bench_roofline.json").read().strip().split("\n")[-1])["roofline"]["mfma_class"]
    open(f"{P}/{R}_x6_step.txt", "w").write(
        "# bf16x6 mode (fp32-class products on the bf16 matrix pipe: three bf16 pieces per operand, six MFMAs per product) in the GAN stage-2 step (B = 64)\n"
        "# per-shape HIP-event timing of every f2g_gemm launch of one step, launch lanes off (F2G_GEMM_REPORT=60 python bench.py --gemm bf16x6 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode);\n"
        "# form 0 = forward / data gradient (gemm_x6_kernel where it applies), 2 = weight gradient (gemm_leanw6_kernel); TFLOP/s = 2*M*N*K / time, fp32-equivalent\n"
        "# the exact-fp32 numbers of the same shapes: profiles/" + R + "_gemm_shapes_fp32.txt\n\n"
        "## default rule (form 0: K >= 2048 and >= 1024 rows; form 2: >= 2048 rows): MFMA class by family " + json.dumps(fam["by_family"]) + "\n"
        + table(f"{G}/x6_shapes.txt", 48) + "\n\n## F2G_X6_MIN_K=32: every eligible GEMM on the six-product kernels (how the rule was found)\n"
        + table(f"{G}/x6_shapes_all.txt", 48) + "\n\n## the laned step (ms_per_step), variants\n" + open(f"{G}/x6_step_variants.txt").read()
        + "\n## tools/x6_gemm_bench.py: the generator's plain GEMMs alone on the chip (image pass of the activation included)\n"
        + "\n".join(l for l in open(f"{G}/x6_gemm_bench.txt").read().split("\n") if "amdgpu.ids" not in l))
if os.path.exists(f"{G}/prof_infer/p_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f"{G}/prof_infer/p_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    # inferences in the trace: 32 launches of the C = 768 block kernel each (8 blocks x 4 steps)
    n768 = sum(int(r["Calls"]) for r in rows if "fused_mlp_kernel<2, 6" in r["Name"]
               or "fused_block_multi_kernel" in r["Name"])
    ninf = n768 // 32 if n768 else 4
    L = ["# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --workload infer4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16",
         f"# BASELINE config 2 (4-step inference, B=64, bf16; one fused launch per layer for the ConvNeXt blocks of all three branches, HIP-graph replay): {tot/ninf/1e6:.2f} ms of kernel time per 4-step inference ({ninf} in the trace: eager warm-up, capture warm-up, replays), launch lanes OFF = every kernel alone on the chip",
         f"{'kernel':66s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'%':>6s}"]
    for r in rows[:24]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:66]
        L.append(f"{n:66s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
    open(f"{P}/{R}_infer4_bf16_kernel_stats.txt", "w").write("\n".join(L) + "\n")
HEAD = {
    "fused_multi.txt": "# tools/fused_multi_bench.py: one layer of the three Fourier branches' ConvNeXt blocks (mel_24k_base, B = 64, bf16) as three launches one after the other / on three streams / as ONE f2g_fused_block_multi launch; then every block alone with the tile height forced (F2G_MLP_RT)\n",
    "pmc_multi.txt": "# tools/pmc_multi.sh: rocprofv3 --pmc passes over tools/fused_multi_bench.py (MODE=multi, MODE=serial); per-kernel means\n",
    "infer4_bf16_timeline_kernels.txt": "# tools/timeline_last.py over the kernel trace of tools/prof_timeline.sh: every kernel of the last graph-replayed bf16 4-step inference (start us, duration us, queue, blocks, kernel)\n",
    "hbm_kernels.txt": "# tools/hbm_kernel_bench.py: the HBM-class ConvNeXt kernels back to back at the mel_24k_base branch shapes (B = 64); algorithmic bytes / HIP-event time against 8 TB/s\n",
    "infer4_bf16_variants.txt": "",
    "pmc_x6_step.txt": "# tools/pmc_x6_step.sh: rocprofv3 --pmc passes over `bench.py --gemm bf16x6 --steps 1 --warmup 1` (launch lanes off); per-kernel means over the launches of gemm_x6_kernel / gemm_leanw6_kernel\n",
    "infer4_bf16_timeline.txt": "# tools/prof_timeline.sh (rocprofv3 --kernel-trace over the HIP-graph replayed bf16 4-step inference, launch lanes ON): overlap statistics of the last inference in the trace (profiled: ~5-15 % slower than unprofiled)\n",
    "wgrad_probe.txt": "", "fused_mlp_bench.txt": "", "fused_mlp_lab.txt": "",
    "conv32_probe.txt": "# tools/conv32_probe.py: direct 32 -> 32 (3,9)/(1,2) MRD band convs at the 45 band shapes of a pass (B = 64: S = 128 forward / weight gradient, S = 64 data gradient); MODE=fp32: exact (persistent forward / data-gradient kernels of round 3, double-buffered weight gradient of round 4), MODE=bf16x6: fp32-class instances on the bf16 pipe (conv32x6.hip, round 4)\n",
    "knockout.txt": "# tools/knockout.py: the laned stage-2 step (B = 64) with components knocked out -- what MPD / MRD / the mel-recon term / the generator cost in the real schedule (step(full) - step(without)); fp32 = exact, bf16x6 = fp32-class mode\n",
    "pmc_lean_fp32.txt": "# tools/pmc_lean_fp32.sh: rocprofv3 --pmc passes over `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode` (exact fp32, launch lanes off)\n",
}
for src, dst in (("conv32_probe.txt", "conv32_probe.txt"), ("knockout.txt", "knockout.txt"),
                 ("pmc_lean_fp32.txt", "pmc_lean_fp32.txt"),
                 ("pmc_x6_step.txt", "x6_step_pmc.txt"), ("lean3_bench.txt", "lean3_bench.txt"), ("conv32_b3_bench.txt", "conv32_split_bench.txt"),
                 ("fused_multi.txt", "fused_multi.txt"), ("pmc_multi.txt", "fused_multi_pmc.txt"),
                 ("infer4_bf16_timeline_kernels.txt", "infer4_bf16_timeline_kernels.txt"),
                 ("hbm_kernels.txt", "hbm_kernels.txt"), ("infer4_bf16_variants.txt", "infer4_bf16_variants.txt"),
                 ("infer4_bf16_timeline.txt", "infer4_bf16_timeline.txt")):
    if os.path.exists(f"{G}/{src}") and os.path.getsize(f"{G}/{src}") > 0:
        body = "\n".join(l for l in open(f"{G}/{src}").read().split("\n") if "amdgpu.ids" not in l)
        open(f"{P}/{R}_{dst}", "w").write(HEAD.get(src, "") + body)
if os.path.exists(f"{G}/bench_infer4_bf16.json"):
    open(f"{P}/{R}_bench_infer4_bf16.json", "w").write(open(f"{G}/bench_infer4_bf16.json").read().strip().split("\n")[-1] + "\n")
txt = open(f"{G}/shapes.txt").read().split("\n")
i = next(k for k, l in enumerate(txt) if l.startswith("form"))
open(f"{P}/{R}_gemm_shapes_fp32.txt", "w").write(
    "# per-shape HIP-event timing of every f2g_gemm launch of ONE GAN stage-2 step (B=64), exact fp32, launch lanes off\n"
    "# (F2G_GEMM_REPORT=80 python bench.py ...); form 0 = forward, 1 = data gradient, 2 = weight gradient; TFLOP/s = 2*M*N*K / time (peak 157.3)\n"
    + "\n".join(txt[i:i + 81]) + "\n")
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = 0.0; n = 0
    for row in csv.DictReader(open(f"{G}/pmcb_{name}/p_counter_collection.csv")):
        if re.search(r"gemm_lean_kernel<(false|0), \d, 0[,>]", row["Kernel_Name"]):   # exact-fp32 instances
            tot += float(row["Counter_Value"]); n += 1
    res[name] = (tot, n)
f, nf = res["FETCH_SIZE"]; w, nw = res["WRITE_SIZE"]
from flow2gan_amd import _lib
json.dump({"source": "F2G_STREAMS=0 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                     "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode; gemm_lean_kernel<0, EP, 0> dispatches (2 steps)",
           "lib_version": _lib.version(), "kernel": "gemm_lean_kernel<0, EP, 0> (exact fp32, all epilogue instances)",
           "launches": nf, "fetch_kib_per_launch": f / nf, "write_kib_per_launch": w / nw,
           "hbm_bytes_per_launch_raw": (f / nf + w / nw) * 1024, "hbm_bytes_per_launch_fetch_x2": (2 * f / nf + w / nw) * 1024,
           "note": "gfx950 FETCH_SIZE under-reports wide coalesced reads by up to 2x (MI355X_MICROARCH.md, HBM); both raw and "
                   "x2-corrected sums given (rocprofv3 reports these counters in KiB)"},
          open(f"{P}/{R}_pmc_gemm_traffic.json", "w"), indent=1)
open(f"{P}/{R}_bench_n1.json", "w").write(open(f"{G}/bench_default.json").read().strip().split("\n")[-1] + "\n")
with open(f"{P}/{R}_other_workloads.jsonl", "w") as fo:
    for f_, cmd in (("bench_stage1", "python bench.py --workload stage1 --steps 10 --warmup 3 --no-cpu-baseline"),
                    ("bench_infer4", "python bench.py --workload infer4 --steps 10 --warmup 3 --no-cpu-baseline"),
                    ("bench_44k", "python bench.py --model mel_44k_128band_512x_base --steps 6 --warmup 2 --no-cpu-baseline"),
                    ("bench_n4", "python bench.py --n-timesteps 4 --steps 4 --warmup 2 --no-cpu-baseline"),
                    ("bench_opt", "python bench.py --optimizer --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode")):
        d = json.loads(open(f"{G}/{f_}.json").read().strip().split("\n")[-1])
        fo.write(json.dumps({"command": cmd, **d}) + "\n")
        print(f_, d["ms_per_step"], d["value"], d.get("fast_mode") and d["fast_mode"].get("ms_per_step"))
d = json.loads(open(f"{P}/{R}_bench_n1.json").read())
print("default", d["ms_per_step"], d["value"], {k: v for k, v in d["roofline"].items() if k not in ("mfma_class", "hbm_class")}, d["cpu_baseline"], d["fast_mode"]["ms_per_step"])
print(open(f"{G}/hbm_kernels.txt").read()); print(open(f"{G}/streaming.txt").read())
