"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the Python mirror keeps the reference's key schema / init / configs, the product path
refuses to run without a GPU (no silent fallback), and the data-parallel gradient exchange is
correct with world_size 2 on gloo."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from flow2gan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "flow2gan_hip.h")).read()
    declared = set(re.findall(r"\b(f2g_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"f2g_stream_t"}
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == declared, set(_lib.EXPORTS) ^ declared
    assert _lib.version().endswith("gfx950")


def test_no_cpu_fallback():
    from flow2gan_amd import _lib, ops
    with pytest.raises(_lib.F2GError):
        ops.fill_(torch.zeros(4), 1.0)
    import flow2gan_amd
    m = flow2gan_amd.MelAudioGenerator(channels=(48, 32, 24), num_layers=(1, 1, 1),
                                       cond_enc_channels=32, cond_enc_num_layers=1,
                                       time_embed_channels=32)
    with pytest.raises(_lib.F2GError):
        m.infer(torch.zeros(1, 100, 8), None, 1)


def _digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().numpy().astype(np.float32).tobytes())
    return h.hexdigest()


def test_generator_init_and_keys_match_reference(golden):
    import flow2gan_amd
    import flow2gan_oracle as O
    from flow2gan_amd.models.config import get_generator_config
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    m = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_24k_base"))
    assert _digest(m.state_dict()) == bytes(g["digest"]).decode()
    assert sum(p.numel() for p in m.parameters()) == 78949542  # reference README / SURVEY
    torch.manual_seed(0)
    o = O.build_generator("mel_44k_128band_512x_base")
    m44 = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_44k_128band_512x_base"))
    assert set(m44.state_dict()) == set(o.state_dict())
    m44.load_state_dict(o.state_dict())  # strict: shapes equal


def test_gan_keys_and_init_match_reference(golden):
    import flow2gan_amd
    import flow2gan_oracle as O
    from flow2gan_amd.models.gan import GAN
    g = golden("tiny_stage2")
    gen = flow2gan_amd.MelAudioGenerator(channels=(48, 32, 24), num_layers=(1, 1, 1),
                                         cond_enc_channels=32, cond_enc_num_layers=1,
                                         time_embed_channels=32)
    torch.manual_seed(int(g["d_seed"]))
    gan = GAN(gen)
    sd = {k: v for k, v in gan.discriminator.state_dict().items() if "spec_fn" not in k}
    assert _digest(sd) == bytes(g["d_digest"]).decode()
    og = O.GAN(O.MelAudioGenerator(channels=(48, 32, 24), num_layers=(1, 1, 1), cond_enc_channels=32,
                                   cond_enc_num_layers=1, time_embed_channels=32))
    assert set(gan.state_dict()) == set(og.state_dict())
    assert sum(p.numel() for p in gan.discriminator.parameters()) == 42503752  # 41.09 M + 1.41 M


def test_configs_and_errors():
    from flow2gan_amd.models import config as C
    import flow2gan_oracle as O
    for name, sub in O.GENERATOR_CONFIGS.items():
        cfg = C.get_generator_config(name)
        for k, v in sub.items():
            assert cfg[k] == v, (name, k)
        assert cfg.sampling_rate == sub["sampling_rate"]  # attribute access like AttributeDict
    assert C.get_gan_config("gan_multi_scale_mel_recon").mel_recon_n_mels == (5, 10, 20, 40, 80, 160, 320)
    with pytest.raises(ValueError):
        C.get_generator_config("nope")
    with pytest.raises(ValueError):
        C.get_gan_config("nope")
    assert C.HF_MODEL_NAMES["libritts-mel-4-step"] == 4


def test_checkpoint_roundtrip_with_ddp_prefix(tmp_path):
    import flow2gan_amd
    from flow2gan_amd.checkpoint import load_checkpoint
    kw = dict(channels=(48, 32, 24), num_layers=(1, 1, 1), cond_enc_channels=32,
              cond_enc_num_layers=1, time_embed_channels=32)
    a = flow2gan_amd.MelAudioGenerator(**kw)
    b = flow2gan_amd.MelAudioGenerator(**kw)
    ck = {"model": {"module." + k: v for k, v in a.state_dict().items()}, "batch_idx_train": 7}
    ck["model"]["module.loss_spec.spectrogram.extra_buffer"] = torch.zeros(3)  # ignored (strict=False)
    torch.save(ck, tmp_path / "epoch-1.pt")
    rest = load_checkpoint(tmp_path / "epoch-1.pt", b)
    assert rest["batch_idx_train"] == 7
    assert _digest(a.state_dict()) == _digest(b.state_dict())
    m, cfg = None, None
    with pytest.raises(AssertionError):
        flow2gan_amd.get_model("mel_24k_base", hf_model_name=None, checkpoint=None)


def _dp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from flow2gan_amd import dist as fdist
    fdist.setup_dist(rank, world, backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300000, 7, 1)]
    params.append(torch.nn.Parameter(torch.zeros(3)))  # no grad: must be skipped
    for i, p in enumerate(params[:-1]):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    moved = fdist.GradReducer(bucket_mb=0.5).reduce(params)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params[:-1]))
    ok = ok and params[-1].grad is None and moved == 4 * (5 + 300000 + 7 + 1)
    # validation totals summed over ranks, keys in sorted order (reference utils.py:318-327)
    tot = fdist.reduce_metrics({"frames": 10.0 * (rank + 1), "loss": 0.5, "a_first": float(rank)})
    ok = ok and tot == {"a_first": 1.0, "frames": 30.0, "loss": 1.0}
    fdist.end_barrier()
    torch.save(ok, os.path.join(out, f"ok{rank}.pt"))
    fdist.cleanup_dist()


def test_grad_reducer_world_size_2_gloo(tmp_path):
    from _mp import run_workers
    run_workers("test_host_side", "_dp_worker", 2, str(tmp_path), timeout=120.0)
    assert torch.load(tmp_path / "ok0.pt") and torch.load(tmp_path / "ok1.pt")


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from flow2gan_amd import dist as fdist
    fdist.setup_dist(rank, world, backend="gloo")
    torch.manual_seed(0)  # same weights on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64),
                              torch.nn.Tanh(), torch.nn.Linear(64, 1))
    unused = torch.nn.Parameter(torch.zeros(5))
    params = list(net.parameters()) + [unused]
    xs = [torch.randn(8, 16, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    # expected: mean over ranks of the local-batch gradients
    want = [torch.zeros_like(p) for p in net.parameters()]
    for x in xs:
        for w, g in zip(want, torch.autograd.grad(net(x).pow(2).mean(), list(net.parameters()))):
            w += g / world
    sent_during_backward = []
    red = fdist.GradReducer(bucket_mb=0.012)  # ~3 buckets
    ok = True
    for it in range(3):  # later passes re-use the arenas and must not accumulate stale values
        red.prepare(params)
        loss = net(xs[rank]).pow(2).mean()
        loss.backward()
        sent_during_backward.append(sum(b.sent for b in red._active.buckets))
        moved = red.finish()
        ok = ok and all(torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7)
                        for p, w in zip(net.parameters(), want))
        ok = ok and unused.grad is None          # used by no rank: None, as without the reducer
        # arenas travel whole: the gradients + one "used" flag per parameter
        ok = ok and moved == 4 * (sum(p.numel() for p in params) + len(params))
    plan = next(iter(red._plans.values()))
    ok = ok and len(red._plans) == 1 and len(plan.buckets) >= 3
    # the first step of a plan records the completion order and sends at finish(); from then on
    # the buckets leave from the autograd hooks, in the order the ranks agreed on
    ok = ok and sent_during_backward[0] == 0 and all(n >= 2 for n in sent_during_backward[1:])
    ok = ok and plan.order is not None and sorted(plan.order) == list(range(len(plan.buckets)))
    # a second backward inside one prepare()/finish() pair must fail loudly, not diverge silently
    red.prepare(params)
    net(xs[rank]).pow(2).mean().backward()
    try:
        net(xs[rank]).pow(2).mean().backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "one backward" in str(e)
    red.finish()
    torch.save(ok, os.path.join(out, f"ov{rank}.pt"))
    fdist.cleanup_dist()


def test_overlapped_grad_reducer_world_size_2_gloo(tmp_path):
    """prepare() / finish(): buckets are exchanged from autograd hooks during backward."""
    from _mp import run_workers
    run_workers("test_host_side", "_overlap_worker", 2, str(tmp_path), timeout=120.0)
    assert torch.load(tmp_path / "ov0.pt") and torch.load(tmp_path / "ov1.pt")


def _sink_worker(rank, world, port, out):
    """Per-lane gradient hand-over (dist._Sink, the mechanism behind fused.deliver_grads): three
    'branch' groups deliver inside one backward, in the order their lanes finish, and a 'rest'
    group arrives through autograd hooks afterwards.  Buckets must leave in exactly that order --
    branch buckets before the rest -- and every rank must end with the mean gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from flow2gan_amd import dist as fdist
    from flow2gan_amd import fused
    fdist.setup_dist(rank, world, backend="gloo")
    torch.manual_seed(0)
    branches = [[torch.nn.Parameter(torch.randn(n)) for n in sizes]
                for sizes in ((40, 7), (30,), (20, 5, 1))]
    rest = [torch.nn.Parameter(torch.randn(9)), torch.nn.Parameter(torch.randn(4))]
    params = rest + [p for b in branches for p in b]          # model order: rest first
    red = fdist.GradReducer(bucket_mb=1.0)
    ok = True
    for it in range(3):
        red.prepare(params, groups=branches)
        sink = fused.GRAD_SINK
        ok = ok and sink is not None
        plan = red._active
        # forward side: branch 0 is used TWICE in this step (two model evaluations), the others once
        keys = [sink.add_use(b) for b in branches]
        keys0b = sink.add_use(branches[0])
        ok = ok and all(k is not None for k in keys) and keys0b == keys[0]
        grad_of = lambda p, use: torch.full_like(p, float(rank + 1) * (use + 1))
        # backward side: evaluation 2 (all three lanes), then evaluation 1 (branch 0 again)
        for i in (2, 1, 0):
            sink.deliver(keys[i], branches[i], [grad_of(p, 0) for p in branches[i]])
        sent_before_second_use = list(plan.sent_order)
        sink.deliver(keys0b, branches[0], [grad_of(p, 1) for p in branches[0]])
        sent_by_branches = list(plan.sent_order)
        # the rest arrives through the ordinary autograd hooks
        loss = sum((p * float(rank + 1)).sum() for p in rest)
        loss.backward()
        order = list(plan.sent_order)
        red.finish()
        ok = ok and fused.GRAD_SINK is None
        b_of = lambda p: plan.bucket_of[id(p)].index
        want_order = [b_of(branches[2][0]), b_of(branches[1][0]), b_of(branches[0][0]), b_of(rest[0])]
        if it == 0:
            # a plan's first step only records the order (and the ranks agree on it at finish())
            ok = ok and order == [] and plan.order == want_order
        else:
            ok = ok and sent_before_second_use == want_order[:2]
            ok = ok and sent_by_branches == want_order[:3]
            ok = ok and order == want_order
        ok = ok and len(set(want_order)) == 4 and len(plan.sent_order) == 4
        for i, b in enumerate(branches):
            want = 1.5 * (3.0 if i == 0 else 1.0)     # mean over ranks of (rank+1) * sum of uses
            ok = ok and all(torch.allclose(p.grad, torch.full_like(p, want)) for p in b)
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in rest)
    torch.save(ok, os.path.join(out, f"sink{rank}.pt"))
    fdist.cleanup_dist()


def test_branch_buckets_leave_before_the_rest_world_size_2_gloo(tmp_path):
    from _mp import run_workers
    run_workers("test_host_side", "_sink_worker", 2, str(tmp_path), timeout=120.0)
    assert torch.load(tmp_path / "sink0.pt") and torch.load(tmp_path / "sink1.pt")


def _unused_worker(rank, world, port, out):
    """The case the reference covers with DDP's find_unused_parameters=True (finetune.py:915): in one
    step a whole group gets no gradient on ONE rank only.  Nothing may hang; every bucket still travels
    on both ranks in the agreed order; the rank without the gradient ends with the ranks' mean (its own
    contribution being zero), and a parameter no rank used comes back as None on both."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from flow2gan_amd import dist as fdist
    from flow2gan_amd import fused
    fdist.setup_dist(rank, world, backend="gloo")
    torch.manual_seed(0)
    branches = [[torch.nn.Parameter(torch.randn(n)) for n in sizes] for sizes in ((40, 7), (30,), (20, 5))]
    rest = [torch.nn.Parameter(torch.randn(9)), torch.nn.Parameter(torch.randn(4))]
    nobody = torch.nn.Parameter(torch.randn(3))
    params = rest + [nobody] + [p for b in branches for p in b]
    red = fdist.GradReducer(bucket_mb=1.0)
    ok = True
    for it in range(4):
        skip = {1} if (it == 2 and rank == 1) else set()      # step 2: rank 1 never runs branch 1
        if it == 3 and rank == 0:
            skip = {0, 2}                                      # step 3: rank 0 runs branch 1 only
        red.prepare(params, groups=branches)
        sink = fused.GRAD_SINK
        plan = red._active
        keys = {i: sink.add_use(branches[i]) for i in range(3) if i not in skip}
        for i in (2, 1, 0):
            if i not in skip:
                sink.deliver(keys[i], branches[i], [torch.full_like(p, float(rank + 1)) for p in branches[i]])
        sum((p * float(rank + 1)).sum() for p in rest).backward()
        moved = red.finish()
        ok = ok and len(plan.sent_order) == len(plan.buckets)           # every bucket, every step
        ok = ok and moved == 4 * (sum(p.numel() for p in params) + len(params))
        ok = ok and nobody.grad is None
        for i, b in enumerate(branches):
            ranks_with = [r for r in range(world)
                          if not ((it == 2 and r == 1 and i == 1) or (it == 3 and r == 0 and i in (0, 2)))]
            want = sum(float(r + 1) for r in ranks_with) / world
            ok = ok and all(p.grad is not None and torch.allclose(p.grad, torch.full_like(p, want)) for p in b)
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in rest)
    # one use with a ticket and one that reaches the parameter through autograd, in one backward:
    # the group's bucket has left when autograd's gradient arrives -- that must fail loudly
    red.prepare(params, groups=branches)
    sink = fused.GRAD_SINK
    keys = [sink.add_use(b) for b in branches]
    for i in (2, 1, 0):
        sink.deliver(keys[i], branches[i], [torch.full_like(p, 1.0) for p in branches[i]])
    try:
        (sum(p.sum() for p in rest) + branches[2][0].sum()).backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "ticket" in str(e)
    red.finish()
    torch.save(ok, os.path.join(out, f"unused{rank}.pt"))
    fdist.cleanup_dist()


def test_group_without_gradient_on_one_rank_world_size_2_gloo(tmp_path):
    from _mp import run_workers
    run_workers("test_host_side", "_unused_worker", 2, str(tmp_path), timeout=120.0)
    assert torch.load(tmp_path / "unused0.pt") and torch.load(tmp_path / "unused1.pt")


def test_gan_stepper_schedule_matches_reference():
    """finetune.py:569-631: D-only until gen_start_batch_idx, then D/G alternation on new batches;
    only the stepped sub-model gets gradients; reference loss weights."""
    from flow2gan_amd.harness import GanLossScales, GanStepper

    class StubGan(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.generator = torch.nn.Linear(1, 1)
            self.discriminator = torch.nn.Linear(1, 1)
            self.calls = []

        def forward(self, cond, audio, audio_lens, n_timesteps, train_disc):
            self.calls.append((train_disc, n_timesteps, float(audio.sum())))
            if train_disc:
                s = self.discriminator(audio[:, :1]).sum()
                return s, 2 * s
            s = self.generator(audio[:, :1]).sum()
            return s, s, s, s, s

    gan = StubGan()
    st = GanStepper(gan, cond_module=lambda a: a, n_timesteps=2, gen_start_batch_idx=3)
    kinds = []
    for i in range(8):
        gan.zero_grad()
        info = st.step(torch.full((2, 4), float(i + 1)), torch.tensor([4, 4]))
        kinds.append(info["train_disc"])
        live = gan.discriminator if info["train_disc"] else gan.generator
        dead = gan.generator if info["train_disc"] else gan.discriminator
        assert all(p.grad is not None for p in live.parameters())
        assert all(p.grad is None for p in dead.parameters())
    assert kinds == [True, True, True, False, True, False, True, False]
    assert [c[2] for c in gan.calls] == [8.0 * (i + 1) for i in range(8)]  # a new batch every step
    assert all(c[1] == 2 for c in gan.calls)
    w = GanLossScales()
    assert (w.disc_loss_mp_scale, w.disc_loss_mr_scale) == (1.0, 0.1)
    assert (w.gen_loss_mp_scale, w.gen_loss_mr_scale, w.feat_map_loss_mp_scale,
            w.feat_map_loss_mr_scale, w.mel_recon_loss_scale) == (1.0, 0.1, 1.0, 0.1, 45.0)


def _small_net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))


def test_checkpoint_roundtrip_and_model_averaging(tmp_path):
    """reference file layout, DDP prefix, running average, range average, plain
    average of files; cross-checked against the reference's own functions where it is present."""
    import copy
    from flow2gan_amd import checkpoint as ck

    cur = _small_net(1)
    avg = copy.deepcopy(cur).to(torch.float64)
    snaps, files = [], []
    period = 2
    for b in range(1, 9):                      # 8 "batches"; average every `period`
        with torch.no_grad():
            for p in cur.parameters():
                p.add_(0.1 * torch.randn_like(p))
        if b % period == 0:
            ck.update_averaged_model({"average_period": period, "batch_idx_train": b}, cur, avg)
            snaps.append({k: v.clone().double() for k, v in cur.state_dict().items()})
            # the running average is the mean of the snapshots taken so far (plus the initial model
            # at weight 0 after the first update: avg = cur * (period / b) + avg * (1 - period / b))
            want = {k: sum(s[k] for s in snaps) / len(snaps) for k in snaps[0]}
            for k, v in avg.state_dict().items():
                assert torch.allclose(v, want[k], rtol=1e-6, atol=1e-7), (b, k)
        if b in (4, 8):
            f = tmp_path / f"checkpoint-{b}.pt"
            ck.save_checkpoint(f, cur, model_avg=avg, params={"batch_idx_train": b, "epoch": 1})
            files.append(f)
    # file layout of the reference
    raw = torch.load(files[0], weights_only=False)
    assert {"model", "optimizer", "scheduler", "grad_scaler", "sampler", "model_avg",
            "batch_idx_train", "epoch"} <= set(raw)
    assert all(v.dtype == torch.float32 for v in raw["model_avg"].values())
    # mean over (4, 8] from the two running averages == mean of snapshots 6 and 8
    rng = ck.average_checkpoints_with_averaged_model(files[0], files[1])
    for k, v in rng.items():
        want = (snaps[2][k] + snaps[3][k]) / 2
        assert torch.allclose(v.double(), want, rtol=1e-5, atol=1e-6), k
    # plain mean of the "model" entries
    plain = ck.average_checkpoints(files)
    a, b_ = (torch.load(f, weights_only=False)["model"] for f in files)
    for k in plain:
        assert torch.allclose(plain[k], (a[k] + b_[k]) / 2)
    # DDP prefix + full round trip incl. model_avg
    raw["model"] = {"module." + k: v for k, v in raw["model"].items()}
    torch.save(raw, tmp_path / "ddp.pt")
    m2, avg2 = _small_net(2), _small_net(3)
    rest = ck.load_checkpoint(tmp_path / "ddp.pt", m2, model_avg=avg2)
    assert rest["batch_idx_train"] == 4 and "model" not in rest and "model_avg" not in rest
    assert all(torch.equal(v, a[k]) for k, v in m2.state_dict().items())
    # the reference's own implementation, when the build container has it
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_shims
    if ref_shims.reference_available():
        ref_shims.install()
        from flow2gan import checkpoint as rck
        ref = rck.average_checkpoints_with_averaged_model(str(files[0]), str(files[1]))
        for k, v in rng.items():
            assert torch.equal(v, ref[k]), k
        ref_plain = rck.average_checkpoints([str(f) for f in files])
        assert all(torch.equal(plain[k], ref_plain[k]) for k in plain)


def test_weight_gradient_split_rule():
    """ops.split_for: blocks = tiles x factor fill rounds of 256 (one block per CU), or of 512 where the
    launch is known to take the two-blocks-per-CU K-major kernel (>= 4096 rows per block) -- the MPD's
    1024-channel layers at B = 64 (tools/micro/leanw_win_probe.py), the generator's shapes unchanged."""
    from flow2gan_amd import ops
    assert ops.split_for(43648, 320) == 4 and ops.split_for(43648, 320, True) == 8      # 2560 blocks = 5 rounds of 512
    assert ops.split_for(39168, 160) == 8 and ops.split_for(39168, 160, True) == 3      # 480 co-resident blocks
    for rows, tiles in ((24064, 27), (6016, 108), (12032, 48), (114048, 20), (341376, 2)):
        s = ops.split_for(rows, tiles)
        assert 1 <= s <= 512 and rows // s >= 16 * 32, (rows, tiles, s)                 # a chunk keeps >= 16 slabs
        s2 = ops.split_for(rows, tiles, True)
        assert s2 == 1 or rows // s2 >= 4096, (rows, tiles, s2)                          # never below the kernel's rule


def test_library_options_and_python_tunables(monkeypatch):
    """Round 6: one options table in the library (f2g_set_option / f2g_get_option; nothing reads the environment
    on a launch path) and one override variable for the Python tunables (F2G_OPTS, flow2gan_amd/_opts.py)."""
    from flow2gan_amd import _lib, _opts
    names = ("lean", "lean_tall", "lean_tap", "lean_wgrad", "x6_tap", "x6_wide", "x6p", "w6t", "deterministic",
             "streamk", "conv2ch_v2", "conv32_v2", "conv32_wgrad_v2", "mlp_rt", "mlp_split", "multi_rt384",
             "multi_rt512", "streamk_min")
    for n in names:
        v = _lib.get_option(n)
        assert _lib.set_option(n, v + 5) == v and _lib.get_option(n) == v + 5
        assert _lib.set_option(n, v) == v + 5 and _lib.get_option(n) == v
    with pytest.raises(_lib.F2GError):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(_lib.F2GError):
        _lib.get_option("x6pr")            # (a removed kernel's switch is not an option any more)
    assert _lib.lib.f2g_set_option(None, 1) != 0
    # Python side: typed like the default, unknown names ignored, booleans from 0 / 1
    monkeypatch.setattr(_opts, "_OPTS", _opts._parse("x6f_min_k=160, fuse_lrelu=3,disc_lanes=0,  x6p=2,junk"))
    assert _opts.opt("x6f_min_k", 384) == 160 and _opts.opt("fuse_lrelu", 0) == 3
    assert _opts.opt("disc_lanes", True) is False and _opts.opt("spec_pad", True) is True
    assert _opts.opt("x6_min_k", 2048) == 2048
    # no per-feature environment switch is left in the package (the kept ones are listed in _opts.py's docstring)
    import glob
    kept = {"F2G_GEMM", "F2G_STREAMS", "F2G_DETERMINISTIC", "F2G_WEIGHT_CACHE", "F2G_LIB_PATH", "F2G_DRYRUN",
            "F2G_DIST_TIMEOUT_S", "F2G_OPTS"}
    for f in glob.glob(os.path.join(ROOT, "flow2gan_amd", "**", "*.py"), recursive=True):
        for m in re.finditer(r'environ(?:\.get)?\(\s*"(F2G_[A-Z0-9_]+)"', open(f).read()):
            assert m.group(1) in kept, (f, m.group(1))
    for f in glob.glob(os.path.join(ROOT, "flow2gan_amd", "csrc", "*.hip")) + \
            glob.glob(os.path.join(ROOT, "flow2gan_amd", "csrc", "*.h")):
        src = open(f).read()
        if not f.endswith("capi.hip"):
            assert "getenv(" not in src, f


def test_weight_batch_levels_operations_by_their_data_dependencies():
    """ops.WeightBatch (the recorder behind rebuild_derived / f2g_multi): an operation's level is one above every
    recorded operation it depends on -- read after write (the image of a transposed copy), write after write,
    write after read (reused memory) --, independent operations share a level.  Host logic only: no launch."""
    from flow2gan_amd import ops
    b = ops.WeightBatch()
    P, A, B_, C_, D_ = 0x1000000, 0x2000000, 0x3000000, 0x4000000, 0x5000000   # parameter, four buffers
    one = dict(n=(1, 1, 1, 1), s=(0, 0, 0, 0), items=10, tensors=())
    b.add(1, A, 4096, P, 4096, **one)              # A = permute(P)                      level 0
    b.add(1, B_, 4096, P + 64, 512, **one)         # B = permute(P): independent          level 0
    b.add(3, C_, 6144, A + 1024, 1024, **one)      # C = split(A): read after write       level 1
    b.add(0, D_, 8192, None, 0, **one)             # fill D                               level 0
    b.add(2, D_ + 256, 1024, C_, 512, **one)       # copy C -> D: RAW on C, WAW on D      level 2
    b.add(0, A + 1536, 64, None, 0, **one)         # overwrite part of A that C read (WAR)  level 2
    b.add(0, A + 2048, 64, None, 0, **one)         # ... a part nobody read: after A's writer only   level 1
    b.add(1, 0x6000000, 128, B_ + 4000, 96, **one)  # reads the tail of B                 level 1
    b.add(1, 0x7000000, 128, B_ + 4096, 96, **one)  # reads just BEHIND B: no dependency  level 0
    assert [o[0] for o in b.ops] == [0, 0, 1, 0, 2, 2, 1, 1, 0]
    # many operations: the lookup stays cheap (it used to scan every recorded range: 155 ms for 600 operations)
    import time
    b = ops.WeightBatch()
    t0 = time.perf_counter()
    for i in range(2000):
        b.add(1, 0x10000000 + i * (1 << 21), 1 << 20, (0x10000000 + (i - 1) * (1 << 21)) if i % 2 else 0x1000 + i * 64,
              1 << 19, **one)
    assert time.perf_counter() - t0 < 0.5
    assert {o[0] for o in b.ops} == {0, 1}
