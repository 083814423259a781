"""Chunked / streaming synthesis (flow2gan_amd/streaming.py; reference infer_dir.py:126-168): the
chunk arithmetic on CPU, GPU parity of a chunked waveform against the oracle running the same
chunking with the same noise, and the HIP-graph replay path."""
import numpy as np
import pytest
import torch

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_chunk_plan_covers_every_sample_once():
    from flow2gan_amd.streaming import chunk_plan
    for frames, chunk in ((70, 20), (100, 100), (101, 50), (7, 3), (300, 64)):
        total = 0
        for fs, fe, lpad, rpad in chunk_plan(frames, chunk, 256):
            assert 0 <= fs < fe <= frames and lpad >= 0
            n = (fe - fs) * 256
            total += len(range(n)[lpad: n - rpad])   # python slicing, as the reference crops
        assert total == frames * 256, (frames, chunk, total)
    # reference arithmetic spelled out for one case (infer_dir.py:146-154)
    assert chunk_plan(70, 20, 256) == [(0, 44, 0, 24 * 256), (0, 64, 20 * 256, 24 * 256),
                                       (16, 70, 24 * 256, 10 * 256), (36, 70, 24 * 256, -10 * 256)]


@pytest.mark.gpu
def test_streaming_infer_matches_oracle_chunking_and_graph_replay(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flow2gan_amd
    import flow2gan_oracle as O
    from flow2gan_amd.streaming import ChunkRunner, chunk_plan, streaming_infer
    g = golden("tiny_forward")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    m = flow2gan_amd.MelAudioGenerator(**TINY)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    o = O.MelAudioGenerator(**TINY)
    o.load_state_dict(sd)
    o.eval()
    gen = torch.Generator().manual_seed(11)
    mel = torch.randn(2, 100, 70, generator=gen) * 2.0 - 5.0
    noises = {}

    def noise_fn(i, B, Tn):
        if i not in noises:
            noises[i] = 0.1 * torch.randn(B, Tn, generator=gen)
        return noises[i].to(DEV)

    got = streaming_infer(m, mel.to(DEV), n_timesteps=2, chunk_size=20, noise_fn=noise_fn)
    # the oracle through the reference's loop
    outs = []
    with torch.no_grad():
        for i, (fs, fe, lpad, rpad) in enumerate(chunk_plan(70, 20, 256)):
            pred = o.infer(mel[:, :, fs:fe], None, 2, True, noise=noises[i])
            outs.append(pred[:, lpad: pred.size(1) - rpad])
    want = torch.cat(outs, dim=-1)
    assert got.shape == want.shape == (2, 70 * 256)
    err = float((got.cpu().double() - want.double()).pow(2).mean().sqrt())
    assert err < 1e-4, err   # north_star waveform tolerance
    # one chunk that holds everything == plain infer
    whole = streaming_infer(m, mel.to(DEV), n_timesteps=2, chunk_size=100,
                            noise_fn=lambda i, B, Tn: noise_fn(100, B, Tn))
    with torch.no_grad():
        plain = m.infer(mel.to(DEV), None, 2, True, noise=noises[100].to(DEV))
    assert torch.equal(whole, plain)
    # HIP-graph replay of the chunk shapes gives the eager result
    runner = ChunkRunner(m, n_timesteps=2)
    for _ in range(2):   # second pass replays every captured shape
        via_graph = streaming_infer(m, mel.to(DEV), n_timesteps=2, chunk_size=20,
                                    noise_fn=noise_fn, runner=runner)
        assert float((via_graph - got).abs().max()) < 1e-6
    assert len(runner.graphs) == 4   # chunk shapes of 44, 64, 54 and 34 frames


@pytest.mark.gpu
def test_graph_replay_with_repeated_interior_chunk_shapes(golden):
    """300 frames in chunks of 64: the three interior chunks share one shape (and one captured
    graph whose static output buffer every replay overwrites) -- the chunks must still be their
    own audio, i.e. the runner's waveform equals the eager one."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flow2gan_amd
    from flow2gan_amd.streaming import ChunkRunner, chunk_plan, streaming_infer
    g = golden("tiny_forward")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    m = flow2gan_amd.MelAudioGenerator(**TINY)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    gen = torch.Generator().manual_seed(5)
    mel = (torch.randn(1, 100, 300, generator=gen) * 2.0 - 5.0).to(DEV)
    plan = chunk_plan(300, 64, 256)
    shapes = [fe - fs for fs, fe, _, _ in plan]
    assert len(plan) == 5 and shapes.count(112) == 3          # 88, 112, 112, 112, 68
    noises = [0.1 * torch.randn(1, n * 256, generator=gen).to(DEV) for n in shapes]
    eager = streaming_infer(m, mel, n_timesteps=1, chunk_size=64, noise_fn=lambda i, B, Tn: noises[i])
    runner = ChunkRunner(m, n_timesteps=1)
    for _ in range(2):
        replay = streaming_infer(m, mel, n_timesteps=1, chunk_size=64,
                                 noise_fn=lambda i, B, Tn: noises[i], runner=runner)
        assert replay.shape == eager.shape == (1, 300 * 256)
        assert float((replay - eager).abs().max()) < 1e-6
    assert len(runner.graphs) == 3
    # distinct chunks really are distinct audio (the aliasing bug produced repeated copies)
    a, b = eager[:, 64 * 256: 128 * 256], eager[:, 128 * 256: 192 * 256]
    assert float((a - b).abs().max()) > 1e-3


@pytest.mark.gpu
def test_graph_replay_follows_weight_changes(golden):
    """A captured chunk graph reads cached re-laid copies of the weights (ops.derived).  After the
    weights change -- in place through torch, through the HIP optimizer's raw pointers, or by
    load_state_dict -- the next call must give the NEW weights' audio (the runner captures the shape
    again), never a replay against the stale copies."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flow2gan_amd
    from flow2gan_amd import optim
    from flow2gan_amd.streaming import ChunkRunner
    g = golden("tiny_forward")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    m = flow2gan_amd.MelAudioGenerator(**TINY)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    gen = torch.Generator().manual_seed(6)
    mel = (torch.randn(1, 100, 48, generator=gen) * 2.0 - 5.0).to(DEV)
    noise = (0.1 * torch.randn(1, 48 * 256, generator=gen)).to(DEV)
    runner = ChunkRunner(m, n_timesteps=2)

    def both():
        with torch.no_grad():
            eager = m.infer(mel, None, 2, True, noise=noise).clone()
        return eager, runner(mel, noise).clone()

    e0, r0 = both()
    assert float((e0 - r0).abs().max()) < 1e-6
    _, r0b = both()                                   # plain replay, nothing changed
    assert float((r0b - r0).abs().max()) < 1e-6 and runner.recaptures == 0
    # (1) an in-place torch write (bumps the parameter's version counter)
    with torch.no_grad():
        m.estimators[0].decoder.blocks[0].pwconv1.weight.mul_(1.5)
        m.estimators[1].decoder.in_proj.weight.mul_(0.5)
    e1, r1 = both()
    assert float((e1 - e0).abs().max()) > 1e-4        # the change is audible
    assert float((e1 - r1).abs().max()) < 1e-6 and runner.recaptures == 1
    # (2) the HIP optimizer (raw-pointer writes, per-parameter epochs)
    opt = optim.ScaledAdam(m.named_parameters(), lr=0.05)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    e2, r2 = both()
    assert float((e2 - e1).abs().max()) > 1e-4
    assert float((e2 - r2).abs().max()) < 1e-6 and runner.recaptures == 2
    # (3) load_state_dict back to the start
    m.load_state_dict(sd)
    e3, r3 = both()
    assert float((e3 - e0).abs().max()) < 1e-6 and float((e3 - r3).abs().max()) < 1e-6
