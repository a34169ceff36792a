#!/usr/bin/env python3
"""conv_ws_kernel vs conv_mfma_kernel on the same inputs (developer tool): the k order is the same, so outputs must be
bit-identical.  DM_CONV_WS is read per launch, so it is toggled in-process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (B, H, cin, cout, mode, extra) in ((64, 96, 64, 64, 0, True), (64, 48, 128, 128, 2, True), (64, 48, 64, 64, 3, False),
                                       (3, 24, 64, 128, 0, True), (5, 48, 192, 64, 0, False), (64, 48, 256, 128, 0, True),
                                       (2, 96, 128, 64, 0, False), (7, 24, 256, 256, 2, False)):
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    wp = K.pack_weight(w)
    bias = torch.randn(cout, device=dev)
    Ho = H * 2 if mode in (2, 3) else H
    kw = {}
    if extra:
        kw = dict(chan_add=torch.randn(B, cout, device=dev), res=torch.randn(B, Ho, Ho, cout, device=dev).to(torch.bfloat16),
                  drop_p=0.1, drop_seed=5)
    outs = []
    for ws in ("0", "1", "4"):
        os.environ["DM_CONV_WS"] = ws
        y = K.conv3x3(x, wp, cout, bias=bias, mode=mode, **kw)
        torch.cuda.synchronize()
        outs.append((K.last_variant(), y.float()))
    for name, y in outs[1:]:
        same = torch.equal(outs[0][1], y)
        ok &= same
        print(f"B={B} H={H} {cin}->{cout} mode={mode} extra={extra}: {outs[0][0]} vs {name}: "
              f"{'bit-identical' if same else 'DIFF max %.3e' % (outs[0][1] - y).abs().max().item()}")
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
