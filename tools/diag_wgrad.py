"""GPU diagnostic: elementwise accuracy of the split weight-gradient kernel against float64 on same-signed operands
(no cancellation: a missing third plane shows up as ~1e-5, fp32-class is ~1e-6 or better)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa: F401,E402
from action_detection_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
cases = [(144, 160, 7, 224, 3, 2), (144, 160, 7, 224, 3, 0), (144, 160, 7, 224, 3, 7), (144, 192, 7, 320, 3, 7),
         (144, 1056, 7, 352, 1, 2), (72, 96, 14, 128, 3, 2), (72, 96, 14, 128, 3, 9), (36, 64, 28, 96, 3, 4),
         (18, 64, 56, 192, 3, 8), (72, 576, 14, 224, 1, 11), (72, 128, 14, 160, 3, 10)]
for (n, cin, h, cout, k, cfg) in cases:
    p = (k - 1) // 2
    x = torch.rand(n, cin, h, h, generator=g) + 0.1
    gy = torch.rand(n, cout, h, h, generator=g) + 0.1
    # float64 reference by unfold (CPU): dW[co][ci][r][s] = sum g * x_shifted
    xu = F.unfold(x.double(), k, padding=p)                      # [n, cin*k*k, h*h]
    ref = torch.einsum("nop,nkp->ok", gy.double().reshape(n, cout, -1), xu).reshape(cout, cin, k, k)
    xd = K.guarded_empty(x.shape, dev)
    xd.copy_(x)
    ws = torch.empty(K.wgrad_x6_workspace_bytes(n, cin, cout, h, h, k, cfg) // 4, device=dev)
    dw, db = torch.empty(cout, cin, k, k, device=dev), torch.empty(cout, device=dev)
    K.conv_wgrad_x6(K.full(gy.to(dev)), K.full(xd), dw, db, k, p, ws, cfg)
    e6 = ((dw.cpu().double() - ref).abs() / ref.abs()).max().item()
    ws2 = torch.empty(K.wgrad_workspace_bytes(n, cin, cout, h, h, k, -1) // 4, device=dev)
    K.conv_wgrad(K.full(gy.to(dev)), K.full(xd), dw, db, k, 1, p, ws2, -1)
    e32 = ((dw.cpu().double() - ref).abs() / ref.abs()).max().item()
    print("wgrad n=%d cin=%d h=%d cout=%d k=%d cfg=%d : x6 max rel %.2e | f32 MFMA kernel %.2e" % (n, cin, h, cout, k, cfg, e6, e32),
          flush=True)
