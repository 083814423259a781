"""Ad-hoc timing of the generator workloads on one GPU (not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flow2gan_amd
from flow2gan_amd.models.config import get_generator_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
what = sys.argv[2] if len(sys.argv) > 2 else "stage1"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda"
torch.manual_seed(0)
m = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(dev)
lm = flow2gan_amd.LogMelSpectrogram().to(dev)
T = 24000
audio = (0.1 * torch.randn(B, T, device=dev)).clamp(-1, 1)
lens = torch.full((B,), T)
mel = lm(audio)

gan = None
if what.startswith("gan"):
    from flow2gan_amd.models.gan import GAN
    m.branch_dropout = 0.0
    gan = GAN(m).to(dev)
    nsteps = int(what[3:] or 1)

def step():
    if gan is not None:
        for p in gan.parameters():
            p.grad = None
        d = gan(mel, audio, lens, nsteps, True)
        (d[0] + 0.1 * d[1]).backward()
        for p in gan.parameters():
            p.grad = None
        ls = gan(mel, audio, lens, nsteps, False)
        sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls)).backward()
    elif what == "stage1":
        m.train()
        loss = m(mel, audio, lens)
        loss.backward()
        for p in m.parameters():
            p.grad = None
    else:
        m.eval()
        with torch.no_grad():
            m.infer(mel, None, 4)

for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"{what} B={B}: {dt*1e3:.1f} ms/step, {B*1.0/dt:.1f} audio-s/s, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
