import sys, os, random
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import flow2gan_amd, flow2gan_oracle as O
import test_hip_gan as tg
g = dict(np.load("tests/golden/tiny_stage2_44k.npz"))
gan = tg.build_gan(flow2gan_amd, g, tg.TINY44)
ogan = tg.build_oracle_gan(g, tg.TINY44)
random.random = lambda: 0.0
T = tg.T
mel, audio, noise = T(g["mel"]).cuda(), T(g["audio"]).cuda(), T(g["noise"]).cuda()
lens = T(g["n1/lens"])
d = gan(mel, audio, lens, 1, True, noise=noise)
gan.zero_grad(); (d[0] + 0.1 * d[1]).backward()
k = "0.discriminators.1.convs.4.bias"
p = dict(gan.discriminator.named_parameters())[k]
ref = T(g["n1/D/g/" + k])
diff = (p.grad.cpu() - ref).abs()
c = int(diff.argmax()); print("channel", c, "err", float(diff[c]), "ref", float(ref[c]), "got", float(p.grad[c]))
with torch.no_grad():
    gan.generator.eval(); ogan.generator.eval()
    fake_h = gan.generator.infer(mel, lens, 1, noise=noise)
    fake_o = ogan.generator.infer(mel.cpu(), lens, 1, noise=noise.cpu())
    print("fake rms diff", float((fake_h.cpu() - fake_o).pow(2).mean().sqrt()))
    _, _, fr_o, ff_o = ogan.discriminator[0](audio.cpu(), fake_o)
    _, _, fr_h, ff_h = gan.discriminator[0](audio, fake_h)
for nm, fo, fh in (("real", fr_o, fr_h), ("fake", ff_o, ff_h)):
    a = fh[1][3].detach().cpu(); b = fo[1][3]
    print(nm, a.shape, b.shape)
    ac, bc = a[:, c], b[:, c]
    idx = bc.abs().flatten().argsort()[:4]
    print("   smallest |oracle| in channel:", [(float(bc.flatten()[i]), float(ac.flatten()[i])) for i in idx])
    print("   flips in channel:", int(((ac > 0) != (bc > 0)).sum()), " max diff", float((ac - bc).abs().max()))
