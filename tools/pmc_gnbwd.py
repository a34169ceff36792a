#!/usr/bin/env python3
"""Launch the GroupNorm(+SiLU) backward at 128 ch @ 48x48, B=64 (one-pass kernel by default; DM_GN_ONEPASS_BWD=0: the two
passes) a few times so that rocprofv3 --pmc passes can attribute HBM traffic to it.  Developer tool, used for profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B, H, C = 64, 48, 128
x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
gam, bet = torch.randn(C, device=dev), torch.randn(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dx = torch.empty_like(x)
_, stats = K.gn_silu_fwd(x, gam, bet, 8)
for _ in range(5):
    K.gn_silu_bwd(x, dy, gam, bet, stats, 8, dg, db, out=dx)
torch.cuda.synchronize()
print("variant:", K.last_variant(), " algorithmic bytes per launch (6n) =", 6 * x.numel(), " two-pass reads+writes (10n) =", 10 * x.numel())
