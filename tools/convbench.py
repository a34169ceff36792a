#!/usr/bin/env python3
"""Per-layer timing of the conv / GEMM kernels at the UNet's own shapes (B=64, STL10 config).
Developer tool: `python tools/convbench.py [--lib path/to/alt/libdm_hip.so] [--B 64]`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", default="")
args = ap.parse_args()

import torch  # noqa: E402
from diffusion_model_amd import _lib  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B = args.B
# name, mode (0 s1, 1 s2, 2 up, 3 zi, 'g' gemm), H(in), Cin, Cout
LAYERS = [
    ("L1.conv 64->64@96", 0, 96, 64, 64),
    ("L1.down s2 64->64", 1, 96, 64, 64),
    ("L2.conv1 64->128@48", 0, 48, 64, 128),
    ("L2.conv2 128->128@48", 0, 48, 128, 128),
    ("L2.down s2 128->128", 1, 48, 128, 128),
    ("L3.conv1 128->256@24", 0, 24, 128, 256),
    ("L3.conv2 256->256@24", 0, 24, 256, 256),
    ("L3.up 256->256 24->48", 2, 24, 256, 256),
    ("L4.conv1 256->128@48", 0, 48, 256, 128),
    ("L4.up 128->128 48->96", 2, 48, 128, 128),
    ("late1 128->64@96", 0, 96, 128, 64),
    ("dgrad late1 64->128@96", 0, 96, 64, 128),
    ("zi dgrad L1.down 64->64 48->96", 3, 48, 64, 64),
    ("gemm L2.qkv 128->384 M=147k", "g", 147456, 128, 384),
    ("gemm L2.skip 64->128 M=147k", "g", 147456, 64, 128),
    ("gemm L3.conv1x1 256->128 M=147k", "g", 147456, 256, 128),
    ("gemm L4.conv1x1 128->64 M=590k", "g", 589824, 128, 64),
    ("gemm L1.to_q 64->64 M=590k", "g", 589824, 64, 64),
    ("gemm L3.qkv 256->768 M=37k", "g", 36864, 256, 768),
]


def timeit(fn, reps):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


for name, mode, H, cin, cout in LAYERS:
    if args.only and args.only not in name:
        continue
    if mode == "g":
        M = H * B // 64
        x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, device=dev) * 0.05
        wp = K.pack_weight(w)
        y = torch.empty(M, cout, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: K.gemm(x, wp, cout, out=y), args.reps)
        fl = 2.0 * M * cin * cout
        by = 2.0 * M * (cin + cout)
    else:
        x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
        wp = K.pack_weight(w)
        Ho = H // 2 if mode == 1 else (2 * H if mode in (2, 3) else H)
        y = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: K.conv3x3(x, wp, cout, mode=mode, out=y), args.reps)
        fl = 2.0 * B * Ho * Ho * cout * cin * 9
        by = 2.0 * B * (H * H * cin + Ho * Ho * cout)
    print(f"{name:36s} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s", flush=True)

# ---- weight gradients (dW += dy^T * patches(x)), same layer shapes
WG = [("wgrad L1.conv 64->64@96", 0, 96, 64, 64), ("wgrad L2.conv2 128->128@48", 0, 48, 128, 128),
      ("wgrad L3.conv2 256->256@24", 0, 24, 256, 256), ("wgrad L3.up 256->256 24->48", 2, 24, 256, 256),
      ("wgrad L4.conv1 256->128@48", 0, 48, 256, 128), ("wgrad L4.up 128->128 48->96", 2, 48, 128, 128),
      ("wgrad late1 128->64@96", 0, 96, 128, 64), ("wgrad L1.down s2 64->64", 1, 96, 64, 64),
      ("wgrad gemm L2.qkv 128->384 M=147k", "g", 147456, 128, 384),
      ("wgrad gemm L4.conv1x1 128->64 M=590k", "g", 589824, 128, 64),
      ("wgrad gemm L1.to_q 64->64 M=590k", "g", 589824, 64, 64),
      ("wgrad gemm L2.skip 64->128 M=147k", "g", 147456, 64, 128),
      ("wgrad gemm L2.out 128->128 M=147k", "g", 147456, 128, 128),
      ("wgrad gemm L4.skip 256->128 M=147k", "g", 147456, 256, 128),
      ("wgrad gemm L3.qkv 256->768 M=37k", "g", 36864, 256, 768),
      ("wgrad gemm L3.out 256->256 M=37k", "g", 36864, 256, 256)]
for name, mode, H, cin, cout in WG:
    if args.only and args.only not in name:
        continue
    if mode == "g":
        M = H * B // 64
        x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(M, cout, device=dev).to(torch.bfloat16)
        dw = torch.zeros(cout, cin, device=dev)
        db = torch.zeros(cout, device=dev)
        t = timeit(lambda: K.wgrad(x, dy, dw, db, taps=1), args.reps)
        fl = 2.0 * M * cin * cout
        by = 2.0 * M * (cin + cout)
    else:
        x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
        Ho = H // 2 if mode == 1 else (2 * H if mode == 2 else H)
        dy = torch.randn(B, Ho, Ho, cout, device=dev).to(torch.bfloat16)
        dw = torch.zeros(cout, cin, 3, 3, device=dev)
        db = torch.zeros(cout, device=dev)
        t = timeit(lambda: K.wgrad(x, dy, dw, db, taps=9, mode=mode), args.reps)
        fl = 2.0 * B * Ho * Ho * cout * cin * 9
        by = 2.0 * B * (H * H * cin + Ho * Ho * cout)
    print(f"{name:36s} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s", flush=True)
