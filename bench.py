#!/usr/bin/env python3
"""Flow2GAN hot-path benchmark on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

Metric (BASELINE.json): audio-seconds/sec of the GAN-stage train step -- one discriminator step +
one generator step of `mel_24k_base`, each on its own synthetic batch of B x 1 s of 24 kHz audio
(reference finetune.py:569-631 alternation, loss weights finetune.py:453-454,478-482), fp32 tensors,
G+D forward/backward + data-parallel gradient all-reduce.  `value` is measured with the fp32-CLASS GEMM
arithmetic (`--gemm bf16x6`, the default: every fp32 operand as three bf16 pieces, the six products with
i + j <= 2 on the bf16 matrix pipe, fp32 accumulate; the GPU parity suite runs in this mode at the
exact-fp32 tolerances); the same step on the exact fp32 MFMA (`--gemm fp32`, the reference's own
arithmetic) is timed in the same run and printed beside it as `exact_fp32`.  One process per GPU; for N > 1 the
driver launches this file through torch.distributed.run and the ranks exchange gradients over
RCCL/xGMI (weak scaling: B per GPU fixed).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel family of one extra, untimed step
with HIP events around every launch) and `cpu_baseline` (the CPU
oracle of the same step timed on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
DTYPES = {"fp32": "f32",
          "bf16x3": "f32 (split-bf16 GEMM, fp32 accumulate)",
          "bf16x6": "f32 (fp32-class products on the bf16 matrix pipe for the long reductions and the direct MRD "
                    "convs: every fp32 operand as three bf16 pieces = 24 significand bits, the six "
                    "v_mfma_f32_32x32x16_bf16 products with i + j <= 2 per fp32 product, fp32 accumulate, error "
                    "<= 2^-23 per product; short reductions and thin layers on the exact fp32 MFMA; activations, "
                    "weights, gradients stay fp32 in HBM)",
          "bf16": "bf16 GEMM operands, fp32 accumulate / activations"}
D_WEIGHTS = (1.0, 0.1)                    # finetune.py:453-454
G_WEIGHTS = (1.0, 0.1, 1.0, 0.1, 45.0)    # finetune.py:478-482


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch (items of 1 s); default 64 (24 kHz) / 32 (44.1 kHz)")
    ap.add_argument("--model", default="mel_24k_base",
                    choices=["mel_24k_base", "mel_44k_128band_512x_base"],
                    help="mel_24k_base = the BASELINE metric; the 44.1 kHz model = config 5")
    ap.add_argument("--n-timesteps", type=int, default=1, help="ODE steps unrolled in the GAN stage")
    ap.add_argument("--workload", default="gan_stage2",
                    choices=["gan_stage2", "stage1", "infer4"],
                    help="gan_stage2 = the BASELINE metric; stage1 / infer4 = configs 3 / 2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--eager-gpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm", default=None, choices=["fp32", "bf16x3", "bf16", "bf16x6"],
                    help="GEMM arithmetic of the headline number.  Default bf16x6 for the training workloads "
                         "(fp32-CLASS products on the bf16 matrix pipe: three bf16 pieces per operand, six MFMAs "
                         "per product, error <= 2^-23 per product; the whole GPU parity suite runs in this mode at "
                         "the exact-fp32 tolerances; the exact-fp32 step is measured in the same run and printed "
                         "as `exact_fp32`), fp32 for infer4.  fp32 = the reference's own arithmetic")
    ap.add_argument("--no-fast-mode", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="infer4: launch every kernel from the host instead of replaying the 4-step "
                         "inference from a captured HIP graph")
    ap.add_argument("--frozen-weights", action="store_true",
                    help="opt out of the per-step invalidation of the derived weight images (transposes, "
                         "three-piece bf16 splits, window-major re-layouts): by default every sub-step ends "
                         "with the invalidation the HIP optimizer performs on the sub-model it stepped "
                         "(ops.bump_weight_epoch), so the timed region rebuilds them as a real train step "
                         "must; with this flag the weights stay frozen and the images are free after warm-up")
    ap.add_argument("--optimizer", action="store_true",
                    help="also run the fused ScaledAdam + Eden2 step inside the timed region "
                         "(a complete train step; the BASELINE metric itself is fwd/bwd)")
    return ap.parse_args()


def synthetic_batch(B, T, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    audio = (0.1 * torch.randn(B, T, generator=g)).clamp_(-1.0, 1.0)
    return audio.to(device)


def main():
    args = parse()
    if args.gemm is None:
        args.gemm = "fp32" if args.workload == "infer4" else "bf16x6"
    if args.eager_gpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload, args.n_timesteps, args.cpu_threads, True,
                                      args.model)),
              flush=True)
        return
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload, args.n_timesteps, args.cpu_threads, False,
                                      args.model)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch through torchrun"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    # F2G_BENCH_ONE_GPU=1 (tests only): every rank drives GPU 0 and the exchange goes through gloo
    # (RCCL refuses two ranks on one device) -- the torchrun path of this file on a 1-GPU box
    one_gpu = os.environ.get("F2G_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import flow2gan_amd
    from flow2gan_amd import dist as fdist
    from flow2gan_amd import ops
    from flow2gan_amd.models.config import get_gan_config, get_generator_config
    from flow2gan_amd.models.gan import GAN

    force_dist = bool(os.environ.get("F2G_FORCE_DIST"))  # exercise RCCL with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        fdist.setup_dist(rank, world, backend="gloo" if one_gpu else "nccl")

    gcfg = get_generator_config(args.model)
    sr = gcfg["sampling_rate"]
    B, T = args.batch or (64 if sr == 24000 else 32), sr
    mel_hop = gcfg["mel_hop_length"]
    torch.manual_seed(1234)  # same weights on every rank (pretrain.py:750, finetune.py:868)
    gen = flow2gan_amd.MelAudioGenerator(**gcfg)
    gen.branch_dropout = 0.0 if args.workload == "gan_stage2" else gen.branch_dropout
    gan = GAN(gen, **get_gan_config("gan_multi_scale_mel_recon")).to(device)
    logmel = flow2gan_amd.LogMelSpectrogram(sr, gcfg["mel_n_fft"], mel_hop, gcfg["n_mels"]).to(device)
    T_inf = T  # config 2: audio_lens=None, output length = mel frames x hop (24064 at 24 kHz)
    reducer = fdist.GradReducer(force=force_dist)
    g_params = list(gan.generator.parameters())
    d_params = list(gan.discriminator.parameters())
    from flow2gan_amd.harness import grad_groups
    g_groups, d_groups = grad_groups(gan, False), grad_groups(gan, True)

    opt_d = opt_g = sch_d = sch_g = None
    if args.optimizer:  # finetune.py:917-921 / pretrain.py:794-799 settings
        from flow2gan_amd.optim import Eden2, ScaledAdam
        opt_g = ScaledAdam(gan.generator.named_parameters(), lr=1e-4, clipping_scale=2.0)
        sch_g = Eden2(opt_g, lr_batches=50000, warmup_start=0.1)
        opt_d = ScaledAdam(gan.discriminator.named_parameters(), lr=1e-4, clipping_scale=2.0)
        sch_d = Eden2(opt_d, lr_batches=50000, warmup_start=0.1)

    FROZEN = [bool(args.frozen_weights)]

    def optimize(opt, sch, params):
        """What follows a sub-step's backward: the optimizer (--optimizer), or at least the invalidation of
        every weight image derived from the sub-model that WOULD have been stepped -- a real train step
        changes those weights, so their transposes / bf16 splits / re-layouts are rebuilt inside the timed
        region (the optimizer itself bumps the same epochs for the tensors it writes)."""
        if opt is not None:
            opt.step()
            sch.step_batch()
        elif not FROZEN[0]:
            ops.bump_weight_epoch(params)
            ops.rebuild_derived(params)      # (batched: what ScaledAdam.step does behind its update)

    audio_d = synthetic_batch(B, T, 1234 + rank, device)
    audio_g = synthetic_batch(B, T, 4321 + rank, device)
    lens = torch.full((B,), T, dtype=torch.int64)
    nts = args.n_timesteps

    # reducer.prepare() zeroes the gradients (views into flat arenas when there is an exchange); a
    # Fourier branch's / period discriminator's bucket starts its all-reduce when its launch lane has
    # finished its backward, the rest from autograd hooks while backward is still running
    COMM = [0, 0]       # bytes handed to the all-reduce, steps

    def step():
        COMM[1] += 1
        if args.workload == "gan_stage2":
            # discriminator step on its batch
            reducer.prepare(d_params, groups=d_groups)
            cond = logmel(audio_d)
            mp, mr = gan(cond, audio_d, lens, nts, True)
            (D_WEIGHTS[0] * mp + D_WEIGHTS[1] * mr).backward()
            COMM[0] += reducer.finish()
            optimize(opt_d, sch_d, d_params)
            # generator step on a new batch
            reducer.prepare(g_params, groups=g_groups)
            cond = logmel(audio_g)
            ls = gan(cond, audio_g, lens, nts, False)
            sum(w * l for w, l in zip(G_WEIGHTS, ls)).backward()
            COMM[0] += reducer.finish()
            optimize(opt_g, sch_g, g_params)
            return 2 * B * (T / sr)
        if args.workload == "stage1":
            gen.train()
            reducer.prepare(g_params, groups=g_groups)
            cond = logmel(audio_g)
            gen(cond, audio_g, lens).backward()
            COMM[0] += reducer.finish()
            optimize(opt_g, sch_g, g_params)
            return B * (T / sr)
        gen.eval()
        with torch.no_grad():
            cond = logmel(audio_g[:, :T_inf])
            if args.no_graph or ops.GEMM_TIMER is not None:   # (per-launch events need eager launches)
                out = gen.infer(cond, None, 4)
            else:
                # the ~600 launches of a 4-step inference replayed from one captured HIP graph per
                # GEMM mode (launch lanes = parallel graph branches): the host no longer bounds it
                out = infer_runner()(cond)
        return B * out.shape[1] / sr

    _runners = {}

    def infer_runner():
        from flow2gan_amd.streaming import ChunkRunner
        key = ops.GEMM_PRECISION
        if key not in _runners:
            _runners[key] = ChunkRunner(gen, n_timesteps=4, clamp_pred=False)
        return _runners[key]

    def barrier():
        if world > 1 or force_dist:
            torch.distributed.barrier()

    HOST_ISSUE = [0.0, 0]

    def timed(nwarm, nsteps):
        for _ in range(nwarm):
            step()
        barrier()
        torch.cuda.synchronize()
        reducer.exposed_comm_ms()       # (drop the warm-up's wait events)
        COMM[0] = COMM[1] = 0
        t0 = time.perf_counter()
        done = 0.0
        for _ in range(nsteps):
            ts = time.perf_counter()
            done += step()
            HOST_ISSUE[0] += time.perf_counter() - ts    # host time to enqueue the step (no sync)
            HOST_ISSUE[1] += 1
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return done, dt

    # headline: --gemm (default bf16x6, fp32 class); the exact-fp32 step follows as `exact_fp32`
    ops.set_gemm_precision(args.gemm)
    reducer.measure = world > 1 or force_dist
    REBUILD0 = [0.0]

    def _mark_rebuild():
        REBUILD0[0] = ops.REBUILD_STATS[0]
    for _ in range(args.warmup):
        step()
    _mark_rebuild()
    audio_s, elapsed = timed(0, args.steps)
    value = world * audio_s / elapsed
    comm = None
    if world > 1 or force_dist:
        # what makes the N > 1 line checkable on its own: the ranks that took part (counted by an
        # all-reduce, not read from the environment), the backend, the bytes every rank handed to
        # the gradient all-reduce per step, and how long the compute stream had to WAIT for the
        # exchange at the end of a backward (event pair around finish()'s wait; max over ranks)
        seen = torch.ones(1, device=device)
        torch.distributed.all_reduce(seen)
        exposed = torch.tensor([reducer.exposed_comm_ms() / max(1, COMM[1])], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(exposed, op=torch.distributed.ReduceOp.MAX)
        comm = {"ranks_seen": int(seen.item()), "backend": torch.distributed.get_backend(),
                "comm_bytes_per_step": int(COMM[0] // max(1, COMM[1])),
                "exposed_comm_ms": round(float(exposed.item()), 3)}
    reducer.measure = False
    host_issue_ms = round(1e3 * HOST_ISSUE[0] / max(1, HOST_ISSUE[1]), 2)
    # host time inside ops.rebuild_derived (Python replay of the weight-image recipes + recording of their launches),
    # per timed step of the headline run
    rebuild_host_ms = round(1e3 * (ops.REBUILD_STATS[0] - REBUILD0[0]) / max(1, HOST_ISSUE[1]), 2)
    def family_table(fam):
        return {k: {"launches": v[0], "tflop": round(v[1] / 1e12, 3), "ms": round(1e3 * v[2], 2),
                    "tflops": round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None}
                for k, v in sorted(fam.items())}

    def event_pass():
        """One extra, untimed step with the launch lanes off and HIP events around every launch."""
        torch.cuda.synchronize()
        tm = ops.GEMM_TIMER = ops.GemmTimer()
        step()
        torch.cuda.synchronize()
        ops.GEMM_TIMER = None
        return tm

    def roofline_of(mode):
        """`roofline` object of one GEMM mode (the precision is already set): dominant kernel family of a
        serialised per-launch HIP-event pass, its PMC traffic, the MFMA class by family, the HBM class."""
        timer = event_pass()
        timer.report_by_epilogue = bool(os.environ.get("F2G_GEMM_REPORT_EPI"))
        n, flops, secs = timer.summary()
        if os.environ.get("F2G_GEMM_REPORT") and rank == 0 and mode == args.gemm:
            print(timer.report(int(os.environ["F2G_GEMM_REPORT"])), file=sys.stderr)
        achieved = flops / secs / 1e12
        fam = timer.by_path()
        hbm = timer.hbm_summary()
        # dominant kernel = the family with the most time in this pass
        dom = max(fam.items(), key=lambda kv: kv[1][2])
        dn, (dl, dfl, dsec) = dom
        # HBM bytes per launch of the dominant kernel come from PMC passes over this same command
        # (profiles/: counters cannot be read from inside the process).  They are only reported when
        # the profile was taken with THIS library version AND describes the family that is dominant in
        # this pass; otherwise null (stale counters would describe different kernels / tiles).
        traffic = None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_x6_traffic.json" if mode == "bf16x6"
                                             else "r*_pmc_gemm_traffic.json")))
        pmc = pmcs[-1] if pmcs else ""            # the newest round's PMC pass (of this mode's dominant family)
        if (args.workload == "gan_stage2" and mode in ("fp32", "bf16x6") and args.model == "mel_24k_base"
                and B == 64 and nts == 1 and os.path.exists(pmc)
                and dn == {"fp32": "lean", "bf16x6": "x6"}[mode]):
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("lib_version") == _lib_version() and pj.get("family", dn) == dn:
                traffic = round(pj["hbm_bytes_per_launch_fetch_x2"])
        alg_bytes = timer.algorithmic_bytes(dn) / max(1, dl)
        # the dense MFMA peak of the arithmetic a family ran in.  bf16x6 mode: the six-product families (the
        # x6 GEMM kernels, the conv32x6 direct convs) run on the bf16 pipe with six MFMAs per product = 2500 / 6
        # TFLOP/s of fp32-class products; what stays on the exact fp32 MFMA is priced against that pipe
        def family_peak(name):
            if mode == "bf16":
                return PEAK_BF16_MFMA_TFLOPS
            if mode == "bf16x6" and (name in ("x6", "x6-thin") or (name == "direct-conv" and ops.CONV32_X6)):
                return PEAK_BF16_MFMA_TFLOPS / 6.0
            if mode == "bf16x3":
                return PEAK_BF16_MFMA_TFLOPS / 3.0
            return PEAK_FP32_MFMA_TFLOPS
        peak = family_peak(dn)
        # (mixed pipes: the time the families would need at their peaks over the time they took)
        class_frac = sum(v[1] / 1e12 / family_peak(k) for k, v in fam.items()) / secs
        ft = family_table(fam)
        for k in ft:
            ft[k]["peak"] = round(family_peak(k), 1)
        return {"bound": "mfma",
                "kernel": {"fused-mlp": "fused_mlp_kernel (whole ConvNeXt block: dwconv7 + BiasNorm in the "
                                        "prologue, pwconv1 -> PReLU -> pwconv2 on bf16 MFMA, z and the "
                                        "hidden activation on chip)",
                           "lean": "gemm_lean_kernel (fp32 MFMA, zero-VALU K loop)",
                           "generic": "gemm_kernel (fp32 MFMA implicit GEMM, generic loaders)",
                           "direct-conv": "conv32 direct kernels (the 32 -> 32 channel MRD band layers)",
                           "lean-streamk": "gemm_lean_kernel (stream-K)",
                           "x6": "gemm_x6p / x6 / x6g / x6f / leanw6 kernels (fp32 class on the bf16 pipe: three bf16 "
                                 "pieces per operand, six MFMAs per product)",
                           "x6-thin": "gemm_x6n_kernel (fp32 class on the bf16 pipe, 128 x 32 tiles: <= 32 output columns)",
                           "narrow": "narrow VALU kernels"}.get(dn, dn),
                # algorithmic FLOPs of the kernel's launches / their summed HIP-event durations
                "achieved": round(dfl / dsec / 1e12, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(dfl / dsec / 1e12 / peak, 4),
                "peak_basis": ("dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32-class product"
                               if peak == PEAK_BF16_MFMA_TFLOPS / 6.0 else
                               "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per split-bf16 product"
                               if peak == PEAK_BF16_MFMA_TFLOPS / 3.0 else
                               "dense bf16 MFMA peak" if peak == PEAK_BF16_MFMA_TFLOPS else
                               "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                # traffic: HBM bytes per launch from the PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE,
                # recorded with THIS build of the library, else null); beside it what one launch must
                # move at least: both operands read once, the result written once
                "traffic": traffic, "algorithmic_bytes_per_launch": round(alg_bytes),
                # (six-product families only) what a bare stream of these MFMAs sustains on real operand data:
                # the spec peak assumes clocks the bf16 pipe does not hold under load
                **({"sustained": {"peak": 291.0, "frac": round(dfl / dsec / 1e12 / 291.0, 4),
                                  "source": "bare six-MFMA stream on N(0,1) bf16 operands, tools/micro/x6_stream_peak.hip -> profiles/r06_x6_stream_peak.txt (1.67 GHz effective: power-limited)"}}
                   if peak == PEAK_BF16_MFMA_TFLOPS / 6.0 else {}),
                "launches_per_step": dl,
                "avg_launch_us": round(1e6 * dsec / dl, 1),
                "share_of_mfma_class_time": round(dsec / secs, 3),
                # every MFMA-class launch of the step (all GEMM families + direct convs); the
                # extra step runs with the launch lanes off (one stream, each kernel alone on the
                # chip), so these durations sum to more than a laned step
                "mfma_class": {"achieved": round(achieved, 2), "frac": round(class_frac, 4),
                               "launches_per_step": n, "ms_per_step_serialised": round(1e3 * secs, 2),
                               "algorithmic_tflop_per_step": round(flops / 1e12, 3),
                               "by_family": ft},
                # HBM-bound kernel class, in-step: algorithmic bytes / HIP-event
                # time per kernel, against the 8 TB/s spec peak
                "hbm_class": {k: {"launches": v[0], "GB": round(v[1] / 1e9, 3), "ms": round(1e3 * v[2], 3),
                                  "achieved_GBps": round(v[1] / v[2] / 1e9, 1),
                                  "frac": round(v[1] / v[2] / 8.0e12, 3)}
                              for k, v in sorted(hbm.items())}}

    roofline = None
    if not args.no_roofline:
        roofline = roofline_of(args.gemm)
        iso = os.path.join(ROOT, "profiles", "hbm_class_isolated.json")
        if os.path.exists(iso):
            # the same HBM-class kernels launched back to back on the step's shapes (tools/hbm_kernel_bench.py
            # on the GPU box; recorded with the library version named inside)
            with open(iso) as f:
                ij = json.load(f)
            if ij.get("lib_version") == _lib_version():
                roofline["hbm_class_isolated"] = ij["kernels"]

    half = max(2, args.steps // 2)
    exact = None
    fast = None
    frozen = None
    if not args.frozen_weights and not args.optimizer and args.workload != "infer4" and not args.no_fast_mode:
        # the same step with the weights frozen (derived weight images free after warm-up): what rounds 1-5
        # reported as `value`; a side field now
        FROZEN[0] = True
        af, ef = timed(2, half)
        FROZEN[0] = False
        frozen = {"value": round(world * af / ef, 2), "unit": "audio-s/s", "ms_per_step": round(1e3 * ef / half, 2),
                  "note": "weights never change, derived weight images cached across steps (--frozen-weights)"}
    if not args.no_fast_mode and args.gemm == "bf16x6":
        # the reference's own arithmetic (exact fp32 MFMA, v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered
        # fmaf chain) measured in the same run, with its own per-launch roofline pass
        ops.set_gemm_precision("fp32")
        a0, e0 = timed(2, half)      # (two warm-up steps: the first one after a mode switch builds that mode's weight copies)
        exact = {"gemm": "exact fp32 MFMA for every GEMM and direct conv", "arithmetic": "fp32",
                 "value": round(world * a0 / e0, 2), "unit": "audio-s/s",
                 "ms_per_step": round(1e3 * e0 / half, 2), "dtype": "f32"}
        if not args.no_roofline:
            r0 = roofline_of("fp32")
            exact["roofline"] = {k: r0[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                    "traffic", "algorithmic_bytes_per_launch",
                                                    "launches_per_step", "avg_launch_us", "mfma_class")}
        ops.set_gemm_precision(args.gemm)
    if not args.no_fast_mode and args.gemm in ("fp32", "bf16x6"):
        # also report the split-bf16 GEMM mode (3 bf16 MFMAs per product, fp32 accumulate): same
        # <= 1e-4 RMS waveform parity (tests/test_hip_generator.py passes in both modes), not fp32-exact
        ops.set_gemm_precision("bf16x3")
        a2, e2 = timed(2, half)
        fast = {"gemm": "split-bf16 (pre-split hi/lo operand images, 3x v_mfma_f32_32x32x16_bf16 per "
                        "product, fp32 accumulate; lean / K-major weight-gradient / direct-conv kernels): "
                        "a THROUGHPUT mode, not a parity mode",
                "value": round(world * a2 / e2, 2), "unit": "audio-s/s",
                "ms_per_step": round(1e3 * e2 / half, 2),
                "parity": "<=1e-4 RMS waveform and losses to 2e-4 vs reference (the golden parity tests "
                          "run in this mode too); per-product error ~2^-16 instead of 2^-24. GRADIENTS "
                          "are held to looser bounds than in exact fp32, because more discriminator "
                          "pixels land on the other side of a leaky-ReLU / L1 / hinge kink: tiny-config "
                          "G-step gradients to 0.15 of the largest gradient among the tensors of their kind and "
                          "0.12 in relative L2 over all of them (fp32: 5e-3 of each tensor's max), "
                          "full-width B=2 gradients to 0.1 (fp32: 1e-2), tests/test_hip_gan.py"}
        if not args.no_roofline:
            tm = event_pass()
            n3, fl3, sec3 = tm.summary()
            fast["mfma_class"] = {
                "launches_per_step": n3, "ms_per_step_serialised": round(1e3 * sec3, 2),
                "achieved_fp32_equivalent_TFLOPs": round(fl3 / sec3 / 1e12, 2),
                # three bf16 MFMAs per product against the dense bf16 peak
                "bf16_mfma_frac": round(3.0 * fl3 / sec3 / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                "by_family": family_table(tm.by_path())}
        if args.gemm == "fp32" and args.workload in ("gan_stage2", "stage1"):
            # fp32-CLASS products on the bf16 matrix pipe (the default headline mode), beside an explicit
            # --gemm fp32 headline
            ops.set_gemm_precision("bf16x6")
            a6, e6 = timed(2, half)
            fast["fp32_class"] = {"gemm": DTYPES["bf16x6"], "value": round(world * a6 / e6, 2),
                                  "unit": "audio-s/s", "ms_per_step": round(1e3 * e6 / half, 2)}
            if not args.no_roofline:
                r6 = roofline_of("bf16x6")
                fast["fp32_class"]["mfma_class"] = r6["mfma_class"]
        if args.workload == "infer4":
            # BASELINE config 2 names bf16 for the generator-only forward: plain bf16 operands,
            # fp32 accumulation and fp32 activations -- a throughput mode, not a parity mode
            ops.set_gemm_precision("bf16")
            a3, e3 = timed(2, half)
            fast["bf16"] = {"gemm": "plain bf16 (1x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate): "
                                    "z and the hidden activation of every block live in HBM as bf16, "
                                    "written by dwnorm / the pwconv1 epilogue; other operands converted",
                            "value": round(world * a3 / e3, 2), "unit": "audio-s/s",
                            "ms_per_step": round(1e3 * e3 / half, 2),
                            "parity": "~1e-3 RMS waveform vs the fp32 path (tests/test_hip_generator.py)"}
        ops.set_gemm_precision(args.gemm)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # N = 1 only: ranks must not wait on it
        # separate process (fresh OpenMP pool, no GPU context) under a hard time limit, so that the
        # reported baseline can never stall the benchmark
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
                                "--workload", args.workload, "--n-timesteps", str(nts),
                                "--cpu-threads", str(args.cpu_threads), "--model", args.model],
                               capture_output=True, text=True, timeout=150)
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            cpu = {"value": None, "unit": "audio-s/s", "cores": None, "kind": "port",
                   "sample": f"cpu baseline did not finish: {type(e).__name__}"}
        # the reference CLIs' own setting is torch.set_num_threads(1) (pretrain.py:894-895,
        # finetune.py:1027-1028): one bounded step at B=1 on a single thread
        try:
            r1 = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
                                 "--workload", args.workload, "--n-timesteps", str(nts),
                                 "--cpu-threads", "1", "--model", args.model],
                                capture_output=True, text=True, timeout=120)
            one = json.loads(r1.stdout.strip().splitlines()[-1])
            cpu["single_thread"] = {"value": one["value"], "cores": 1, "sample": one["sample"]}
        except Exception as e:  # noqa: BLE001
            cpu["single_thread"] = {"value": None, "cores": 1,
                                    "sample": f"did not finish: {type(e).__name__}"}

    if rank == 0:
        line = {
            "metric": f"audio-seconds/sec (train step, G+D fwd/bwd) {args.model}",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "host_issue_ms_per_step": host_issue_ms,
            "rebuild_host_ms_per_step": rebuild_host_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.gemm],
            # which arithmetic `value` was measured in; `exact_fp32` carries the reference's own (exact fp32
            # MFMA) from the same run -- compare like with like across rounds
            "arithmetic": {"fp32": "fp32", "bf16x6": "fp32-class (bf16x6)", "bf16x3": "split-bf16 (bf16x3)",
                           "bf16": "bf16"}[args.gemm],
            "data": "synthetic (0.1*randn clipped, seeded per rank); seeded random-init weights",
            "config": {"workload": args.model + " " + {
                           "gan_stage2": "GAN stage-2 train step: D-step + G-step, "
                                         "each on its own batch (MPD+MRD+FM+multi-scale mel), "
                                         "fwd+bwd+grad all-reduce" + (" + optimizer step" if args.optimizer
                                                                    else ", no optimizer (metric is fwd/bwd)"),
                           "stage1": "flow-matching stage-1 fwd+bwd",
                           "infer4": "4-step Euler inference" + ("" if args.no_graph else
                                                                 " replayed from a captured HIP graph")}[args.workload],
                       "per_gpu_batch": B, "seconds_per_item": T / sr,
                       "n_timesteps": 4 if args.workload == "infer4" else nts,
                       "gan": "gan_multi_scale_mel_recon", "parallelism": f"dp{world}",
                       "optimizer": "ScaledAdam + Eden2 (fused HIP)" if args.optimizer else "none",
                       "weights": ("written by the optimizer every sub-step" if args.optimizer else
                                   "frozen: derived weight images cached across steps" if args.frozen_weights
                                   or args.workload == "infer4" else
                                   "derived weight images (transposes, bf16 splits, re-layouts) invalidated "
                                   "after every sub-step, as an optimizer step would: rebuilt inside the timed "
                                   "region")},
            "roofline": roofline, "exact_fp32": exact, "frozen_weights": frozen, "cpu_baseline": cpu,
            "fast_mode": fast,
        }
        if comm is not None:
            line.update(comm)
        # RCCL prints its version banner through C stdio (flushed at exit): push it out first so
        # that the JSON line is the last line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        fdist.cleanup_dist()


def _lib_version() -> str:
    from flow2gan_amd import _lib
    return _lib.version()


def cpu_baseline(workload: str, nts: int, threads: int = 0, eager_gpu: bool = False,
                 model: str = "mel_24k_base"):
    """The CPU oracle (oracle/flow2gan_oracle.py, a port validated against the reference) on the
    host cores, bounded: B=8 x 1 s, one warm-up + timed steps for >= 10 s (at most 6).

    eager_gpu=True (hidden flag --eager-gpu-baseline-only, never part of the default run) times the
    same PyTorch restatement as eager ROCm kernels (MIOpen / rocBLAS / hipFFT) on the GPU at the
    full B=64: what the reference's own code path costs on this MI355X, for DESIGN.md's table."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow2gan_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads if threads > 0 else max(1, min(avail // 2, 32))  # physical cores, capped
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    gen = O.build_generator(model)
    gc = O.GENERATOR_CONFIGS[model]
    sr, hop = gc["sampling_rate"], gc["mel_hop_length"]
    B, T = ((64 if sr == 24000 else 32) if eager_gpu else (1 if threads == 1 else 8)), sr
    dev = torch.device("cuda" if eager_gpu else "cpu")
    audio = (0.1 * torch.randn(B, T)).clamp_(-1, 1).to(dev)
    lens = torch.full((B,), T, device=dev)
    lm = O.LogMelSpectrogram(sr, gc["mel_n_fft"], hop, gc["n_mels"]).to(dev)
    gen = gen.to(dev)
    sync = torch.cuda.synchronize if eager_gpu else (lambda: None)
    if workload == "gan_stage2":
        gen.branch_dropout = 0.0
        gan = O.GAN(gen).to(dev)

        def step():
            gan.zero_grad()
            mp, mr = gan(lm(audio), audio, lens, nts, True)
            (D_WEIGHTS[0] * mp + D_WEIGHTS[1] * mr).backward()
            gan.zero_grad()
            ls = gan(lm(audio), audio, lens, nts, False)
            sum(w * l for w, l in zip(G_WEIGHTS, ls)).backward()
            return 2 * B * 1.0
    elif workload == "stage1":
        gen.train()

        def step():
            gen.zero_grad()
            gen(lm(audio), audio, lens).backward()
            return B * 1.0
    else:
        gen.eval()

        def step():
            with torch.no_grad():
                gen.infer(lm(audio), None, 4)
            return B * (1 + T // hop) * hop / sr
    for _ in range(3 if eager_gpu else (0 if threads == 1 else 1)):
        step()
    sync()
    t0 = time.perf_counter()
    done, nstep = 0.0, 0
    while nstep < (1 if threads == 1 else 6) and (time.perf_counter() - t0 < 10.0 or nstep == 0):
        done += step()
        sync()
        nstep += 1
    dt = time.perf_counter() - t0
    if eager_gpu:
        return {"value": round(done / dt, 3), "unit": "audio-s/s", "ms_per_step": round(1e3 * dt / nstep, 2),
                "kind": "PyTorch eager restatement of the reference on this GPU (MIOpen/rocBLAS/hipFFT)",
                "sample": f"B={B} x 1 s, 3 warm-up + {nstep} timed step(s), fp32"}
    return {"value": round(done / dt, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch CPU fp32 restatement), B={B} x 1 s, {0 if threads == 1 else 1} warm-up + {nstep} timed "
                      f"step(s) of the same workload ({dt:.1f} s)"}


if __name__ == "__main__":
    main()
