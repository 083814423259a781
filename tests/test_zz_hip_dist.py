"""Data-parallel semantics of the real HIP path with two ranks: both processes drive the same MI355X
(the test box has one GPU; RCCL refuses two ranks on one device, so the exchange goes through gloo,
which accepts device tensors) through GanStepper + the overlapped GradReducer.  Every rank must end
up with the mean over ranks of the per-rank gradients that a single process computes."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.multiproc]

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)


def _build():
    import flow2gan_amd
    from flow2gan_amd.models.gan import GAN
    torch.manual_seed(1234)                       # same weights on every rank
    gen = flow2gan_amd.MelAudioGenerator(**TINY)
    gen.branch_dropout = 0.0
    gan = GAN(gen).to("cuda")
    logmel = flow2gan_amd.LogMelSpectrogram(24000, 1024, 256, 100).to("cuda")
    return gan, logmel


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return (0.1 * torch.randn(2, 6000, generator=g)).clamp(-1, 1).to("cuda"), torch.tensor([6000, 6000])


def _steps(gan, logmel, rank, reducer):
    """One D-step and one G-step (fixed generator noise); returns the gradients of both."""
    import random
    from flow2gan_amd.harness import GanStepper
    random.seed(0)
    audio, lens = _batch(rank)
    st = GanStepper(gan, logmel, n_timesteps=1, gen_start_batch_idx=1, reducer=reducer)
    out = {}
    for name in ("D", "G"):
        torch.manual_seed(7)                      # the step's own randn draw for the ODE noise
        gan.zero_grad()
        info = st.step(audio, lens)
        assert info["train_disc"] == (name == "D")
        sub = gan.discriminator if name == "D" else gan.generator
        out[name] = {k: p.grad.detach().cpu().clone() for k, p in sub.named_parameters()}
    return out


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    from flow2gan_amd import dist as fdist
    import torch.distributed as dist
    fdist.setup_dist(rank, world, backend="gloo")     # bounded rendezvous (F2G_DIST_TIMEOUT_S)
    gan, logmel = _build()
    red = fdist.GradReducer(bucket_mb=8.0)
    got = _steps(gan, logmel, rank, red)
    assert len(red._plans) == 2 and max(len(p.buckets) for p in red._plans.values()) >= 2
    torch.save(got, os.path.join(outdir, f"rank{rank}.pt"))
    totals = fdist.reduce_metrics({"b": float(rank + 1), "a": 10.0}, device=torch.device("cuda"))
    assert totals == {"a": 20.0, "b": 3.0}, totals
    fdist.end_barrier()
    dist.destroy_process_group()


def _single(rank, world, port, outdir):
    """What rank `rank` computes on its own (no exchange), in a fresh process with the workers'
    environment (F2G_DETERMINISTIC=1: no library-chosen split-K / stream-K)."""
    torch.cuda.set_device(0)
    from flow2gan_amd import dist as fdist
    gan, logmel = _build()
    torch.save(_steps(gan, logmel, rank, fdist.GradReducer()), os.path.join(outdir, f"single{rank}.pt"))


def test_two_ranks_average_the_single_rank_gradients(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from _mp import gloo_rendezvous_works, run_workers
    from flow2gan_amd import dist as fdist
    pre = tmp_path / "preflight"
    pre.mkdir()
    if not gloo_rendezvous_works(str(pre)):
        pytest.skip("two CPU-only processes cannot rendezvous over gloo/127.0.0.1 on this box")
    # fresh worker processes, free port, killed (and the test failed) after 150 s
    run_workers("test_zz_hip_dist", "_worker", 2, str(tmp_path), timeout=150.0,
                env={"F2G_DETERMINISTIC": "1"})
    # what each rank computes on its own (no exchange): fresh processes, same environment
    run_workers("test_zz_hip_dist", "_single", 2, str(tmp_path), timeout=150.0,
                env={"F2G_DETERMINISTIC": "1"})
    singles = [torch.load(tmp_path / f"single{r}.pt") for r in range(2)]
    for r in range(2):
        got = torch.load(tmp_path / f"rank{r}.pt")
        for name in ("D", "G"):
            worst = 0.0
            for k, v in got[name].items():
                want = 0.5 * (singles[0][name][k] + singles[1][name][k])
                err = float((v - want).abs().max()) / (float(want.abs().max()) + 1e-9)
                worst = max(worst, err)
            # F2G_DETERMINISTIC=1 in the workers (no library-chosen split-K); what remains is the
            # atomic accumulation order of the weight gradients
            assert worst < 2e-3, (r, name, worst)


def test_bench_step_under_torchrun_with_two_ranks(tmp_path):
    """bench.py's own step() through `python -m torch.distributed.run --nproc-per-node 2` (both
    ranks on this box's single GPU, gloo exchange): the exact launch line the driver uses on the
    8-GPU node must not run for the first time there.  Checks the JSON contract of rank 0's line."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    import subprocess
    import sys
    from _mp import ROOT, free_port, gloo_rendezvous_works
    pre = tmp_path / "preflight"
    pre.mkdir()
    if not gloo_rendezvous_works(str(pre)):
        pytest.skip("two CPU-only processes cannot rendezvous over gloo/127.0.0.1 on this box")
    env = dict(os.environ, F2G_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo", F2G_DIST_TIMEOUT_S="90",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline", "--no-roofline", "--no-fast-mode"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # the N > 1 line checks itself: ranks counted by an all-reduce, backend, bytes per step, exposed wait
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo"
    # the default line = the headline arithmetic (round 5: fp32-class products on the bf16 pipe)
    assert "six" in d["dtype"] and d["exact_fp32"] is None          # (--no-fast-mode: no second arithmetic timed)
    # bytes per step handed to the all-reduce: the live sub-model only -- D-step 170.015 MB (216 discriminator
    # tensors), G-step 315.798 MB (425 generator tensors) -- plus one "used" flag per tensor, instead of the
    # reference's 485.8 MB twice through DDP-over-GAN (finetune.py:913-915)
    assert d["comm_bytes_per_step"] == 170015008 + 315798168 + 4 * (216 + 425), d["comm_bytes_per_step"]
    assert d["exposed_comm_ms"] >= 0.0
    # whole-job aggregate: both ranks' audio over the max-over-ranks time
    assert abs(d["value"] - 2 * 2 * 4 * 1.0 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]


class _SpyReducer:
    """GradReducer that remembers, per step, which buckets had left BEFORE finish() was called."""

    def __new__(cls, *a, **kw):
        from flow2gan_amd import dist as fdist

        class Spy(fdist.GradReducer):
            def __init__(self, *a, **kw):
                super().__init__(*a, **kw)
                self.log = []

            def finish(self):
                plan = self._active
                if plan is not None:
                    self.log.append((list(plan.sent_order), len(plan.buckets), plan))
                return super().finish()

        return Spy(*a, **kw)


def _rccl_single(rank, world, port, outdir):
    """ONE rank through the real RCCL backend (what F2G_FORCE_DIST=1 does in bench.py), launch lanes
    ON: arenas, per-lane hand-over, asynchronous all-reduce on the communication stream, the
    stream ordering between up to seven lanes and that stream -- against the plain path."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    import torch.distributed as dist
    from flow2gan_amd import dist as fdist
    from flow2gan_amd import fused, ops
    from flow2gan_amd.harness import grad_groups
    assert ops.CONCURRENT
    fdist.setup_dist(0, 1, backend="nccl")
    gan, logmel = _build()
    plain = _steps(gan, logmel, 0, fdist.GradReducer())            # world 1, not forced: no exchange
    spy = _SpyReducer(bucket_mb=8.0, force=True)
    first = _steps(gan, logmel, 0, spy)       # a plan's first step records the completion order and
    got = _steps(gan, logmel, 0, spy)         # sends at finish(); from the second on buckets leave early
    torch.cuda.synchronize()
    assert fused.GRAD_SINK is None
    assert spy.log[0][0] == [] and spy.log[1][0] == [], spy.log[:2]
    for name in ("D", "G"):
        for k, v in got[name].items():
            assert float((v - first[name][k]).abs().max()) <= 2e-4 * (float(v.abs().max()) + 1e-9), (name, k)
    (d_order, d_n, d_plan), (g_order, g_n, g_plan) = spy.log[2:]
    # every bucket had left before finish(): from a lane's hand-over or an autograd hook
    assert sorted(d_order) == list(range(d_n)) and sorted(g_order) == list(range(g_n)), (d_order, g_order)
    assert d_n >= 6 and g_n >= 4
    # G-step: the three branch groups' buckets leave before anything of the rest (cond encoder,
    # condition paths), which closes last
    branch_ids = {id(p) for grp in grad_groups(gan, False) for p in grp}
    is_branch = {b.index: all(id(p) in branch_ids for p in b.params) for b in g_plan.buckets}
    assert any(is_branch.values()) and not all(is_branch.values())
    first_rest = min(i for i, bi in enumerate(g_order) if not is_branch[bi])
    assert all(is_branch[bi] for bi in g_order[:first_rest])
    assert not any(is_branch[bi] for bi in g_order[first_rest:]), g_order
    worst = {}
    for name in ("D", "G"):
        w = 0.0
        for k, v in got[name].items():
            want = plain[name][k]
            w = max(w, float((v - want).abs().max()) / (float(want.abs().max()) + 1e-9))
        worst[name] = w
    torch.save(worst, os.path.join(outdir, "rccl_single.pt"))
    dist.destroy_process_group()


def test_single_rank_rccl_exchange_with_lanes_matches_plain_path(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from _mp import run_workers
    run_workers("test_zz_hip_dist", "_rccl_single", 1, str(tmp_path), timeout=200.0)
    worst = torch.load(tmp_path / "rccl_single.pt")
    # same kernels, same lanes; what differs is atomics' order and the arena accumulation
    assert worst["D"] < 2e-4 and worst["G"] < 2e-4, worst


@pytest.mark.gpu
def test_half_precision_arena_keeps_the_dtype_agnostic_path():
    """`f2g_bucket_arm` / `f2g_scale` are float kernels: an arena of bf16 / fp16 parameters must not take
    them (they would write 4 * n bytes + flags at float offsets into a 2 * (n + nflags)-byte buffer)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from flow2gan_amd import dist as fdist
    guard = torch.full((4096,), 7.0, device="cuda")      # (lands behind the arena in the caching allocator's block)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        ps = [torch.nn.Parameter(torch.ones(33, 5, device="cuda", dtype=dt)),
              torch.nn.Parameter(torch.ones(7, device="cuda", dtype=dt))]
        plan = fdist._Plan(ps, 1 << 30)
        assert len(plan.buckets) == 1
        b = plan.buckets[0]
        assert b.hip_fp32 == (dt == torch.float32)
        b.flat.fill_(3.0)
        plan.arm()
        torch.cuda.synchronize()
        assert b.flat.dtype == dt and b.flat.numel() == 33 * 5 + 7 + 2
        assert float(b.flat[:b.n].float().abs().max()) == 0.0
        assert b.flat[b.n:].float().tolist() == [1.0, 1.0]
        assert all(p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape for p, v in zip(b.params, b.views))
        st = b.flag_staging()
        assert st.dtype == dt and st.is_pinned() and b.flag_staging() is st
    assert float(guard.min()) == 7.0 and float(guard.max()) == 7.0
