#!/usr/bin/env python3
"""Launch the bench's dominant kernel (nearest-x2 + conv3x3, L4.upsample.conv shape, B=64) a few times so
that a rocprofv3 --pmc pass can attribute HBM traffic to it (developer tool, used for profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B, H, cin, cout = 64, 48, 128, 128
x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
wp = K.pack_weight(w)
bias = torch.zeros(cout, device=dev)
y = torch.empty(B, 2 * H, 2 * H, cout, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    K.conv3x3(x, wp, cout, bias=bias, mode=2, out=y)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", x.numel() * 2 + y.numel() * 2 + wp.numel() * wp.element_size())
