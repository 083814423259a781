import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import torch
import flow2gan_oracle as O
from flow2gan_amd.models.discriminators import MultiResolutionDiscriminator
for Tn in (6001, 11025, 11264, 44100):
    torch.manual_seed(5)
    x = 0.1 * torch.randn(2, Tn); x[1] *= 3.0
    y = 0.1 * torch.randn(2, Tn)
    torch.manual_seed(9)
    do = O.MultiResolutionDiscriminator(); dh = MultiResolutionDiscriminator()
    dh.load_state_dict(do.state_dict(), strict=False); dh = dh.cuda()
    with torch.no_grad():
        sr_o, sf_o, fr_o, ff_o = do(x, y)
    sr_h, sf_h, fr_h, ff_h = dh(x.cuda(), y.cuda())
    for i, (fa, fb) in enumerate(zip(fr_h, fr_o)):
        errs = []
        for j, (a, b) in enumerate(zip(fa, fb)):
            d = (a.detach().cpu().double() - b.double()).abs()
            errs.append("%.1e" % (float(d.max()) / float(b.abs().max())))
        print(Tn, "sub", i, "shapes", tuple(fb[0].shape), " ".join(errs))
