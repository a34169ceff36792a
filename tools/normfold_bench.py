#!/usr/bin/env python3
"""Folded GroupNorm (dm_gn_prepare + norm_in) against the materialised path (dm_gn_silu_fwd + conv / wgrad on its output)
at the ResBlock shapes of the STL10 UNet, B=64.  Developer tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for name, H, cin, cout in (("L1 64->64@96", 96, 64, 64), ("L2.conv1 64->128@48", 48, 64, 128), ("L2.conv2 128->128@48", 48, 128, 128),
                           ("L3.conv2 256->256@24", 24, 256, 256), ("L4.conv1 256->128@48", 48, 256, 128)):
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, cout, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    wp = K.pack_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.03)
    y = torch.empty(B, H, H, cout, dtype=torch.bfloat16, device=dev)
    dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
    g, _ = K.gn_silu_fwd(x, gamma, beta, 8)
    _, nin = K.gn_prepare(x, gamma, beta, 8)
    t_gn = timeit(lambda: K.gn_silu_fwd(x, gamma, beta, 8))
    t_prep = timeit(lambda: K.gn_prepare(x, gamma, beta, 8))
    t_c0 = timeit(lambda: K.conv3x3(g, wp, cout, out=y))
    v0 = K.last_variant()
    t_c1 = timeit(lambda: K.conv3x3(x, wp, cout, out=y, norm_in=nin))
    v1 = K.last_variant()
    t_w0 = timeit(lambda: K.wgrad(g, dy, dw, db))
    t_w1 = timeit(lambda: K.wgrad(x, dy, dw, db, norm_in=nin))
    print(f"{name:24s} gn_silu_fwd {t_gn:6.1f}  gn_prepare {t_prep:6.1f} | conv {t_c0:6.1f} -> folded {t_c1:6.1f} | "
          f"wgrad {t_w0:6.1f} -> folded {t_w1:6.1f} | fwd pair {t_gn + t_c0:6.1f} -> {t_prep + t_c1:6.1f} us   [{v0} / {v1}]", flush=True)
