"""Diagnostic (round 6): the 3b block-input launch (256 -> 256 at 28 x 28, rows >= 64 stored 256 channels further up, rows >= 192 raw) on
every tile config at the bench batch: output and recorded maximum against an fp32 matmul."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa
from action_detection_amd import _lib, kernels as K, planes as P

dev = torch.device("cuda:0")
action_detection_amd.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
g = torch.Generator().manual_seed(2)
for (cin, h, cout, split, gap, raw_from, ctot_all) in [(256, 28, 256, 64, 256, 192, 512), (192, 28, 224, 64, 192, 192, 416)]:
    x = torch.randn(n, cin, h, h, generator=g).clamp(min=0).to(dev)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(dev)
    ctot = split + gap + (cout - split)
    scale = (torch.rand(ctot, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
    z = torch.einsum("nchw,oc->nohw", x.double(), w[:, :, 0, 0].double())
    ref = torch.zeros(n, ctot_all, h, h, dtype=torch.float64, device=dev)
    for m in range(cout):
        d = m if m < split else m + gap
        ref[:, d] = torch.relu(z[:, m] * scale[d].double() + shift[m].double()) if m < raw_from else z[:, m]
    xp = P.from_f32(x)
    wp = K.pack_weights_multi([([w], 0)], x6=True)[0]
    for tile in ([7, 11, 0] if len(sys.argv) > 2 else range(12)):
        y = P.PlaneTensor(n, ctot_all, h, h, dev).zero_()
        if len(sys.argv) > 2:
            y.data.fill_(float(sys.argv[2]))
        for _ in range(2):
            P.conv_fwd(P.pfull(xp), wp, scale, shift, P.PSlice(y, 0, cout), 1, 1, 1, 0, 0, True, tile, raw_from=raw_from, row_split=split, row_gap=gap)
            am = y.amax.item()
            y.pool.update()
        got = P.to_f32(y)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        print("cin %d tile %2d  rel err %.3g  amax %.6g (true %.6g)" % (cin, tile, err, am, ref.abs().max().item()), flush=True)
        if len(sys.argv) > 2 and tile != 0:
            written = [d for d in range(ctot_all) if (d < split or (split + gap <= d < ctot))]
            bad = ((got - ref).abs() > 1e-5 * ref.abs().max())
            bad[:, [d for d in range(ctot_all) if d not in written]] = False
            idx = bad.nonzero()
            print("   bad elements %d of %d; channels %s" % (idx.shape[0], bad.numel(), sorted(set(idx[:, 1].tolist()))[:40]))
            print("   images %s rows %s cols %s" % (sorted(set(idx[:, 0].tolist()))[:10], sorted(set(idx[:, 2].tolist()))[:30], sorted(set(idx[:, 3].tolist()))[:30]))
            hi = y.data[0].float(); lo = y.data[1].float()
            for (nn, c, hh, ww) in idx[:6].tolist():
                print("   [n %d c %d h %d w %d] got %.6f ref %.6f  hi %.4f lo %.6f scale %.1f" % (nn, c, hh, ww, got[nn, c, hh, ww].item(), ref[nn, c, hh, ww].item(),
                      hi[nn, c // 8, hh * h + ww, c % 8].item(), lo[nn, c // 8, hh * h + ww, c % 8].item(), y.scale.item()))
