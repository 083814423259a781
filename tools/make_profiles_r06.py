"""Turn the raw outputs of tools/collect_r06.sh (gpurun_out/r06/) into profiles/r06_* (+ profiles/hbm_class_isolated.json)."""
import csv, json, os, re, sys
G, P, R = "gpurun_out/r06", "profiles", "r06"
sys.path.insert(0, os.getcwd())


def stats(path, out, header, steps=3):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    L = [header, f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/steps:.1f} ms per step ({steps} steps in the trace)",
         f"{'kernel':66s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}"]
    for r in rows[:56]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:66]
        L.append(f"{n:66s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} "
                 f"{float(r['MinNs'])/1e3:8.1f} {float(r['MaxNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
    g = [r for r in rows if re.search(r"gemm_kernel|gemm_lean|narrow_|conv32_|conv33_|gemm_x6", r["Name"])]
    gc = sum(int(r["Calls"]) for r in g); gt = sum(float(r["TotalDurationNs"]) for r in g)
    L.append(f"# all gemm_lean / gemm_x6* / gemm_kernel<...> / narrow_* / conv32_* / conv33_* dispatches (= the MFMA-class launches "
             f"bench.py times; the thin first / last discriminator layers -- conv2ch_*, convpost_*, mpd0_*, mpdpost_* -- are HBM class): "
             f"{gc} calls, {gt/1e6:.1f} ms, average {gt/gc/1e3:.1f} us, {100*gt/tot:.1f} % of kernel time")
    open(out, "w").write("\n".join(L) + "\n")
    print(L[1]); print(L[-1])


CMD = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
stats(f"{G}/prof_serial/p_kernel_stats.csv", f"{P}/{R}_gan_stage2_kernel_stats.txt",
      f"# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- {CMD}\n"
      "# (mel_24k_base GAN stage-2 step, B=64 x 1 s per D-step and per G-step, the HEADLINE arithmetic: bf16x6 = fp32-class products "
      "on the bf16 matrix pipe, 1 x MI355X)\n"
      "# launch lanes OFF: every kernel alone on the chip -- the same condition as bench.py's roofline pass, whose per-launch average must agree")
stats(f"{G}/prof_lanes/p_kernel_stats.csv", f"{P}/{R}_gan_stage2_kernel_stats_lanes.txt",
      f"# rocprofv3 --kernel-trace --stats -- {CMD}\n"
      "# default mode: launch lanes ON (up to 7 HIP streams): kernel durations overlap and stretch, their sum exceeds the wall time of a step")
stats(f"{G}/prof_fp32/p_kernel_stats.csv", f"{P}/{R}_exact_fp32_kernel_stats.txt",
      f"# F2G_STREAMS=0 rocprofv3 --kernel-trace --stats -- {CMD} --gemm fp32\n"
      "# the same step on the exact fp32 MFMA (the reference's own arithmetic; `exact_fp32` of the bench line), launch lanes OFF")


def table(path, n):
    t = [l for l in open(path).read().split("\n") if "amdgpu.ids" not in l]
    i = next(k for k, l in enumerate(t) if l.startswith("form"))
    return "\n".join(t[i:i + n + 1])


open(f"{P}/{R}_gemm_shapes_x6.txt", "w").write(
    "# per-shape HIP-event timing of every MFMA-class launch of ONE GAN stage-2 step (B=64) in the headline arithmetic (bf16x6), launch lanes off\n"
    "# (F2G_GEMM_REPORT=90 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode); form 0 = forward, 1 = data gradient, "
    "2 = weight gradient; TFLOP/s = 2*M*N*K / time, fp32-equivalent; family = the kernel family bench.py counts the launch in\n"
    + table(f"{G}/shapes_x6.txt", 90) + "\n")
open(f"{P}/{R}_gemm_shapes_fp32.txt", "w").write(
    "# the same table on the exact fp32 MFMA (--gemm fp32; peak 157.3 TFLOP/s)\n" + table(f"{G}/shapes_fp32.txt", 90) + "\n")
open(f"{P}/{R}_bench_n1.json", "w").write(open(f"{G}/bench_default.json").read().strip().split("\n")[-1] + "\n")
with open(f"{P}/{R}_other_workloads.jsonl", "w") as fo:
    for f_, cmd in (("bench_stage1", "python bench.py --workload stage1 --steps 10 --warmup 3 --no-cpu-baseline"),
                    ("bench_infer4", "python bench.py --workload infer4 --steps 10 --warmup 3 --no-cpu-baseline"),
                    ("bench_44k", "python bench.py --model mel_44k_128band_512x_base --steps 6 --warmup 2 --no-cpu-baseline"),
                    ("bench_n4", "python bench.py --n-timesteps 4 --steps 4 --warmup 2 --no-cpu-baseline"),
                    ("bench_opt", "python bench.py --optimizer --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode"),
                    ("bench_frozen", "python bench.py --frozen-weights --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode"),
                    ("bench_stage1_frozen", "python bench.py --workload stage1 --frozen-weights --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode")):
        d = json.loads(open(f"{G}/{f_}.json").read().strip().split("\n")[-1])
        fo.write(json.dumps({"command": cmd, **d}) + "\n")
        print(f_, d["ms_per_step"], d["value"], (d.get("exact_fp32") or {}).get("ms_per_step"), d.get("fast_mode") and d["fast_mode"].get("ms_per_step"))
open(f"{P}/{R}_bench_infer4_bf16.json", "w").write(open(f"{G}/bench_infer4_bf16.json").read().strip().split("\n")[-1] + "\n")
HEAD = {
    "hbm_kernels.txt": "# tools/hbm_kernel_bench.py: the HBM-class ConvNeXt kernels back to back at the mel_24k_base branch shapes (B = 64); algorithmic bytes / HIP-event time against 8 TB/s (the json of the same run: profiles/hbm_class_isolated.json = roofline.hbm_class_isolated of the bench line)\n",
    "knockout.txt": "# tools/knockout.py: the laned stage-2 step (B = 64) with components knocked out -- what MPD / MRD / the mel-recon term / the generator cost in the real schedule (step(full) - step(without)); bf16x6 = the headline fp32-class mode, fp32 = exact\n",
    "conv32_probe.txt": "# tools/conv32_probe.py: direct 32 -> 32 (3,9)/(1,2) MRD band convs at the 45 band shapes of a pass (B = 64: S = 128 forward / weight gradient, S = 64 data gradient); MODE=fp32: exact, MODE=bf16x6: fp32-class instances on the bf16 pipe (conv32x6.hip)\n",
    "pmc_busy_x6.txt": "# tools/pmc_busy.sh r06_x6: one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAIT_INST_ANY, ...) + kernel trace over `bench.py --steps 1 --warmup 1` in the headline arithmetic, launch lanes off; GHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / duration, i.e. 8 x the effective clock; mfma_busy = matrix-pipe busy share of the kernel's life (profiles/r06_pmc_busy_x6_before.txt: the same pass at the start of the round)\n",
    "pmc_busy_fp32.txt": "# tools/pmc_busy.sh r06_fp32 --gemm fp32: the same pass over the exact-fp32 step\n",
}
for src in HEAD:
    if os.path.exists(f"{G}/{src}") and os.path.getsize(f"{G}/{src}") > 0:
        body = "\n".join(l for l in open(f"{G}/{src}").read().split("\n") if "amdgpu.ids" not in l)
        open(f"{P}/{R}_{src}", "w").write(HEAD[src] + body)
HEAD2 = {
    "steady_state_kernels.txt": "# tools/dbg/copybuffer_context.sh: kernel trace (launch lanes off) of `bench.py --steps 1 --warmup 1 --no-roofline --no-fast-mode`; the SECOND step only = steady state with the weights invalidated after every sub-step: kernels per step by name, and what surrounds every __amd_rocclr_copyBuffer (the 160 / step of the round-5 stats were the cold first step: module .to(device), first-use tables; a steady-state step has the six host-to-device uploads of the per-branch frame-length vectors)\n",
    "shapes_stage1_epi.txt": "# F2G_GEMM_REPORT=40 F2G_GEMM_REPORT_EPI=1 python bench.py --workload stage1 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-mode: the generator's GEMMs by shape AND epilogue kind (prelu2 = PReLU with both outputs, dprelu = PReLU backward + two column sums, res = residual * gamma, atomic = weight gradient), launch lanes off\n",
}
for src, head in HEAD2.items():
    if os.path.exists(f"{G}/{src}"):
        body = "\n".join(l for l in open(f"{G}/{src}").read().split("\n") if "amdgpu.ids" not in l)
        open(f"{P}/{R}_{src}", "w").write(head + body)
for src in ("pmc_x6_traffic.json", "pmc_gemm_traffic.json"):
    open(f"{P}/{R}_{src}", "w").write(open(f"{G}/{src}").read())
open(f"{P}/hbm_class_isolated.json", "w").write(open(f"{G}/hbm_class_isolated.json").read())
d = json.loads(open(f"{P}/{R}_bench_n1.json").read())
print("default", d["ms_per_step"], d["value"], {k: v for k, v in d["roofline"].items() if k not in ("mfma_class", "hbm_class", "hbm_class_isolated")})
print("exact_fp32", {k: v for k, v in d["exact_fp32"].items() if k != "roofline"}, d["exact_fp32"]["roofline"]["frac"])
print(d["cpu_baseline"], d["fast_mode"]["ms_per_step"])
