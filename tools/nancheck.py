#!/usr/bin/env python3
"""Find the first kernel call whose output holds a NaN (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K
from diffusion_model_amd import unet as U

def wrap(name):
    f = getattr(K, name)
    def g(*a, **kw):
        r = f(*a, **kw)
        outs = [t for t in (list(a) + list(kw.values()) + [r]) if isinstance(t, torch.Tensor) and t.is_floating_point()]
        for t in outs:
            if torch.isnan(t).any():
                shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)]
                print("NaN after", name, shapes, {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}, flush=True)
                raise SystemExit(1)
        return r
    setattr(K, name, g)
    if hasattr(U, "K"):
        pass
for n in ["gemm", "conv3x3", "wgrad", "gn_silu_fwd", "gn_silu_bwd", "attn_fwd", "attn_bwd"]:
    if hasattr(K, n):
        wrap(n)
torch.manual_seed(0)
m = U.UNET(input_channels=3, output_channels=3, context_dim=512, dropout_prob=0.0).cuda().eval()
with torch.no_grad():
    m.output_conv.weight.normal_(0, 0.05)
x = torch.randn(2, 3, 48, 48, device="cuda", requires_grad=True)
t = torch.tensor([10, 400], device="cuda")
import numpy as np
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/unet_stl10_init_noctx.npz"))
x = torch.tensor(g["x"], device="cuda", requires_grad=True); t = torch.tensor(g["t"])
print(x.shape, t)
y = m(x, t, None)
y.backward(torch.randn_like(y))
print("ok", torch.isnan(x.grad).any().item())
