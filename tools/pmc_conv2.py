#!/usr/bin/env python3
"""A few launches of several conv / wgrad instantiations at their B=64 UNet shapes, for rocprofv3 --pmc SQ passes
(LDS bank conflicts, MFMA busy cycles, wait cycles).  Developer tool; results go under profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
B = 64
for (H, cin, cout, mode) in ((48, 128, 128, 2), (96, 64, 64, 0), (48, 128, 128, 0), (96, 64, 64, 1)):
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    wp = K.pack_weight(w)
    bias = torch.zeros(cout, device=dev)
    for _ in range(3):
        y = K.conv3x3(x, wp, cout, bias=bias, mode=mode)
    print(K.last_variant())
    if mode == 0:
        dw = torch.zeros(cout, cin, 3, 3, device=dev)
        for _ in range(3):
            K.wgrad(x, torch.randn_like(y), dw)
        print(K.last_variant())
torch.cuda.synchronize()
