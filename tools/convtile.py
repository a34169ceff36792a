#!/usr/bin/env python3
"""Sweep of dm_conv3x3's tile candidates (16x16 / 8x24 / 8x8 output pixels x 64 / 128 couts) over the UNet's own
conv shapes at B=64, interleaved rounds in one process (the numbers pick_tile()'s cost model is fitted to).
Developer tool: `python tools/convtile.py [--B 64] [--rounds 5]`; uses the DM_CONV_FORCE override of dm_conv3x3."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="")
args = ap.parse_args()

import torch  # noqa: E402
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B = args.B
# name, mode, H(in), Cin, Cout, pool2
LAYERS = [
    ("input_conv 64p->64@96", 0, 96, 64, 64, 0),
    ("L1.conv 64->64@96 (+dgrad, late2)", 0, 96, 64, 64, 0),
    ("late1 128->64@96", 0, 96, 128, 64, 0),
    ("dgrad late1 64->128@96", 0, 96, 64, 128, 0),
    ("L4.up 128->128 48->96", 2, 48, 128, 128, 0),
    ("dgrad L4.up pool2 128->128@96", 0, 96, 128, 128, 1),
    ("zi dgrad L1.down 64->64 48->96", 3, 48, 64, 64, 0),
    ("L2.conv1 64->128@48", 0, 48, 64, 128, 0),
    ("dgrad L2.conv1 128->64@48", 0, 48, 128, 64, 0),
    ("L2.conv2 128->128@48 (+dgrad, L4.conv2)", 0, 48, 128, 128, 0),
    ("L4.conv1 256->128@48", 0, 48, 256, 128, 0),
    ("dgrad L4.conv1 128->256@48", 0, 48, 128, 256, 0),
    ("L3.up 256->256 24->48", 2, 24, 256, 256, 0),
    ("dgrad L3.up pool2 256->256@48", 0, 48, 256, 256, 1),
    ("zi dgrad L2.down 128->128 24->48", 3, 24, 128, 128, 0),
    ("L3.conv1 128->256@24", 0, 24, 128, 256, 0),
    ("dgrad L3.conv1 256->128@24", 0, 24, 256, 128, 0),
    ("L3.conv2 256->256@24 (+dgrad)", 0, 24, 256, 256, 0),
]
TILES = {0: "16x16", 1: "8x24", 2: "8x8"}


def timeit(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


for name, mode, H, cin, cout, pool2 in LAYERS:
    if args.only and args.only not in name:
        continue
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    wp = K.pack_weight(w)
    Ho = 2 * H if mode in (2, 3) else H
    if pool2:
        Ho //= 2
    y = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
    fn = lambda: K.conv3x3(x, wp, cout, mode=mode, out=y, pool2=bool(pool2))  # noqa: E731
    cands = [("auto", None)]
    for t in ((0,) if pool2 else (0, 1, 2)):
        for nt in ((4, 8) if cout % 128 == 0 else (4,)):
            cands.append((f"{TILES[t]}/nt{nt}", f"{t},{nt}"))
    best = {c: 1e9 for c, _ in cands}
    variants = {}
    for r in range(args.rounds + 1):
        for c, force in cands:
            if force is None:
                os.environ.pop("DM_CONV_FORCE", None)
            else:
                os.environ["DM_CONV_FORCE"] = force
            fn()
            variants[c] = K.last_variant()
            t = timeit(fn, args.reps)
            if r > 0:
                best[c] = min(best[c], t)
    os.environ.pop("DM_CONV_FORCE", None)
    fl = 2.0 * B * (Ho * (2 if pool2 else 1)) ** 2 * cout * cin * 9
    line = "  ".join(f"{c}={best[c]*1e6:6.1f}" for c, _ in cands)
    bc = min(best, key=best.get)
    print(f"{name:42s} {line}   best {bc} {fl/best[bc]/1e12:6.0f} TF/s (auto {fl/best['auto']/1e12:6.0f}; {variants['auto']})", flush=True)
