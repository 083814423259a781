"""Latency of chunked synthesis at batch 1 (mel_24k_base, 4 Euler steps): eager launches vs
HIP-graph replay (flow2gan_amd/streaming.py:ChunkRunner)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flow2gan_amd
from flow2gan_amd.models.config import get_generator_config
from flow2gan_amd.streaming import ChunkRunner, streaming_infer

dev = "cuda"
torch.manual_seed(0)
m = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(dev).eval()
frames, chunk, nts = 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 100, 4
mel = torch.randn(1, 100, frames, device=dev) * 2 - 5
secs = frames * 256 / 24000
for name, runner in (("eager", None), ("hip-graph", ChunkRunner(m, n_timesteps=nts))):
    for _ in range(2):
        streaming_infer(m, mel, n_timesteps=nts, chunk_size=chunk, runner=runner)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        streaming_infer(m, mel, n_timesteps=nts, chunk_size=chunk, runner=runner)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nch = (frames + chunk - 1) // chunk
    print(f"{name:10s} chunk={chunk} frames: {dt*1e3:8.1f} ms for {secs:.1f} s of audio "
          f"({secs/dt:6.1f} x real time), {dt/nch*1e3:6.2f} ms per chunk of {chunk*256/24000:.2f} s")
