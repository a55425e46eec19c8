"""Diagnostic (round 6): one eager forward of the BN-Inception backbone at the bench batch, printing amax * scale of every activation
tensor after the first pass (before calibration settles) and which launch produced a non-finite maximum."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa
from action_detection_amd import planes_exec as E, planes as P, kernels as K
from action_detection_amd.bninception import BNInception
from action_detection_amd.synthetic import init_backbone_synthetic

dev = torch.device("cuda:0")
action_detection_amd.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
torch.manual_seed(1234)
net = BNInception()
init_backbone_synthetic(net)
net.to(dev).eval()
x = (torch.randint(0, 256, (n, 3, 224, 224)).float() - 110.0).to(dev)
orig = P.conv_fwd
log = []
def spy(xs, wp, scale, shift, y, kh, kw, stride, ph, pw, relu=True, tile_cfg=-1, raw_from=0, row_split=0, row_gap=0):
    orig(xs, wp, scale, shift, y, kh, kw, stride, ph, pw, relu, tile_cfg, raw_from, row_split, row_gap)
    torch.cuda.synchronize()
    a = float(y.t.amax.item())
    log.append((a, y.c0, y.c, y.hw, xs.c, kh, stride, tile_cfg, raw_from, row_split, row_gap))
    if a != a or a == float("inf"):
        print("NON-FINITE amax after conv_fwd: dst c0 %d c %d hw %s cin %d k %d s %d tile %d raw_from %d split %d gap %d" %
              (y.c0, y.c, y.hw, xs.c, kh, stride, tile_cfg, raw_from, row_split, row_gap), flush=True)
        d = P.to_f32(y)
        print("   stored data finite: %s, max |stored| %g; x finite %s" % (bool(torch.isfinite(d).all()), d.abs().max().item(),
              bool(torch.isfinite(P.to_f32(xs)).all())), flush=True)
        raise SystemExit(1)
P.conv_fwd = spy
E.P.conv_fwd = spy
try:
    with torch.no_grad():
        f = net.features(x)
    print("forward ok; feat finite", bool(torch.isfinite(f).all()), "launches", len(log))
except SystemExit:
    pass
except Exception as e:
    print("EXC", str(e)[:400])
