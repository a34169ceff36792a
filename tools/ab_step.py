#!/usr/bin/env python3
"""Same-box A/B of the training step with two builds of libdm_hip.so (developer tool):
python tools/ab_step.py alt/libdm_prev.so [steps]  -> runs bench.py's training leg with each library, twice, alternating."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
alt = os.path.abspath(sys.argv[1])
steps = sys.argv[2] if len(sys.argv) > 2 else "20"
code = ("import sys, os; sys.path.insert(0, %r); sys.argv = ['bench.py', '--steps', %r, '--warmup', '3', '--no-cpu-baseline', "
        "'--no-latent', '--no-elided', '--no-mnist', '--no-roofline', '--sample-images', '0']; "
        "from diffusion_model_amd import _lib; lp = os.environ.get('DM_AB_LIB'); "
        "_lib.LIB_PATH = lp or _lib.LIB_PATH; import bench; bench.main()") % (root, steps)
for rnd in range(2):
    for name, lib in (("alt", alt), ("cur", "")):
        env = dict(os.environ, DM_AB_LIB=lib)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"{name}: {d['value']:.0f} img/s  {d['ms_per_step']:.3f} ms/step  loss {d['final_loss']:.6f}", flush=True)
        except Exception:
            print(name, "FAILED", out.stderr[-400:], flush=True)
