import sys, os, random
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import flow2gan_amd
import test_hip_gan as tg
g = dict(np.load("tests/golden/tiny_stage2_44k.npz"))
gan = tg.build_gan(flow2gan_amd, g, tg.TINY44)
random.random = lambda: 0.0
T = tg.T
mel, audio, noise = T(g["mel"]).cuda(), T(g["audio"]).cuda(), T(g["noise"]).cuda()
for tag, n in (("n1", 1), ("n2", 2)):
    lens = T(g[f"{tag}/lens"])
    d = gan(mel, audio, lens, n, True, noise=noise)
    print(tag, [float(v) for v in d], g[f"{tag}/D/losses"])
    gan.zero_grad()
    (1.0 * d[0] + 0.1 * d[1]).backward()
    rows = []
    for k, p in gan.discriminator.named_parameters():
        key = f"{tag}/D/g/{k}"
        if key in g:
            ref = T(g[key]).double()
            diff = (p.grad.cpu().double() - ref).abs()
            rows.append((float(diff.max()) / float(ref.abs().max() + 1e-12), k, float(diff.max()), float(ref.abs().max()),
                         int((diff > 0.1 * diff.max()).sum()), diff.numel()))
    rows = [r for r in rows if r[3] > 1e-8]
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("   %.2e %-50s maxerr %.2e refmax %.2e  n(>0.1max)=%d/%d" % r)
