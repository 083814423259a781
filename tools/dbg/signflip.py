import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import flow2gan_amd, flow2gan_oracle as O
import test_hip_gan as tg
g = dict(np.load("tests/golden/tiny_stage2_44k.npz"))
gan = tg.build_gan(flow2gan_amd, g, tg.TINY44)
og = O.MelAudioGenerator(**tg.TINY44)
torch.manual_seed(int(g["d_seed"]))
ogan = O.GAN(og)
audio = tg.T(g["audio"])
for di, name in ((0, "MPD"), (1, "MRD")):
    with torch.no_grad():
        sr_o, _, fr_o, _ = ogan.discriminator[di](audio, audio)
    sr_h, _, fr_h, _ = gan.discriminator[di](audio.cuda(), audio.cuda())
    for i, (fa, fb) in enumerate(zip(fr_h, fr_o)):
        for j, (a, b) in enumerate(zip(fa, fb)):
            a = a.detach().cpu()
            flips = int(((a > 0) != (b > 0)).sum())
            if flips:
                idx = ((a > 0) != (b > 0)).nonzero()[0]
                print(name, "sub", i, "fmap", j, "sign flips:", flips, "of", a.numel(), "values", float(a[tuple(idx)]), float(b[tuple(idx)]))
print("done")
