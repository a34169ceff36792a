"""A few launches of the L2 self-attention kernels (B=64, N=2304, hd=16, dropout 0.1) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B, N, C = 64, 2304, 128
qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16)
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
for _ in range(3):
    o, lse = K.attn_fwd(q, k, v, 8, drop_p=0.1, drop_seed=7)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, torch.randn_like(o), 8, drop_p=0.1, drop_seed=7)
torch.cuda.synchronize()
