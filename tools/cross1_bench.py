#!/usr/bin/env python3
"""Timing of the one-token cross-attention output side at the UNet's shapes (B=64): fused dm_cross1_* against the
three-kernel form it replaces.  Developer tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for N, C in ((9216, 64), (2304, 128), (576, 256)):
    g = torch.Generator().manual_seed(0)
    v = torch.randn(B, 1, C, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(C, C, generator=g) * C ** -0.5).to(dev)
    res, dy = (torch.randn(B, N, C, generator=g).to(dev).to(torch.bfloat16) for _ in range(2))
    wp, wpt = K.pack_weight(w), K.pack_weight(w, transpose=True)
    dw = torch.zeros(C, C, device=dev)
    for p in (0.1, 0.0):
        def old_f():
            o = K.attn_onekey_fwd(v, N, 8, drop_p=p, drop_seed=3)
            return K.gemm(o, wp, C, res=res), o
        _, o = old_f()

        def old_b():
            d_o = K.gemm(dy, wpt, C)
            K.wgrad(o, dy, dw, None, taps=1)
            return K.attn_onekey_bwd(v, o, d_o, 8, drop_p=p, drop_seed=3)
        print(f"N={N} C={C} drop {p}: fwd {timeit(old_f):7.1f} -> {timeit(lambda: K.cross1_fwd(v, w, N, 8, res=res, drop_p=p, drop_seed=3)):7.1f} us   "
              f"bwd {timeit(old_b):7.1f} -> {timeit(lambda: K.cross1_bwd(dy, v, w, 8, dw_out=dw, drop_p=p, drop_seed=3)):7.1f} us", flush=True)
