#!/usr/bin/env python3
"""Launch the HBM-bound kernels (GroupNorm+SiLU fwd/bwd at the L1 shape, the streaming GEMM, the sampler step) a few
times so that rocprofv3 --pmc passes can attribute HBM traffic to them (developer tool, used for profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B, H, C = 64, 96, 64
x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
gam, bet = torch.randn(C, device=dev), torch.randn(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
xg = torch.randn(B * H * H, 128, device=dev).to(torch.bfloat16)
wp = K.pack_weight(torch.randn(64, 128, device=dev) * 0.05)
S = 88
z, vu, vc, nz = (torch.randn(S, 3, 96, 96, device=dev) for _ in range(4))
coef = torch.rand(S, 8, device=dev)
for _ in range(5):
    y, stats = K.gn_silu_fwd(x, gam, bet, 8)
    K.gn_silu_bwd(x, dy, gam, bet, stats, 8, dg, db)
    K.gemm(xg, wp, 64)
    K.cfg_ddpm_step(z, vu, vc, nz, coef)
torch.cuda.synchronize()
n = x.numel()
print("algorithmic bytes per launch: gn fwd (stats 2n, apply 4n) =", 2 * n, 4 * n, " gn bwd (reduce 4n, apply 6n) =", 4 * n, 6 * n,
      " gemm 128->64 =", xg.numel() * 2 + xg.shape[0] * 64 * 2, " cfg_ddpm_step =", 20 * z.numel())
