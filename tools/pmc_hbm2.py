#!/usr/bin/env python3
"""Launch the one-pass kernels of round 2's last session a few times so that rocprofv3 --pmc passes can attribute HBM
traffic to them: GroupNorm forward in one pass (256 ch @ 48x48, B=64) and the one-token cross-attention output side
(dm_cross1_fwd / dm_cross1_bwd at 9216 pixels x 64 channels, B=64, dropout 0.1).  Developer tool, used for profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B = 64
x = torch.randn(B, 48, 48, 256, device=dev).to(torch.bfloat16)
gam, bet = torch.randn(256, device=dev), torch.randn(256, device=dev)
N, C = 9216, 64
v = torch.randn(B, 1, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, device=dev) * 0.1
res, dy = (torch.randn(B, N, C, device=dev).to(torch.bfloat16) for _ in range(2))
dw = torch.zeros(C, C, device=dev)
y = torch.empty_like(x)
for _ in range(5):
    K.gn_silu_fwd(x, gam, bet, 8, out=y)
    K.cross1_fwd(v, w, N, 8, res=res, drop_p=0.1, drop_seed=3)
    K.cross1_bwd(dy, v, w, 8, dw_out=dw, drop_p=0.1, drop_seed=3)
torch.cuda.synchronize()
print("algorithmic bytes per launch: gn one-pass fwd (4n) =", 4 * x.numel(), " cross1 fwd (4n) =", 4 * res.numel(),
      " cross1 bwd (2n) =", 2 * dy.numel())
