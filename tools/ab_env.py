#!/usr/bin/env python3
"""Same-box A/B of bench.py legs under an environment switch (developer tool):
python tools/ab_env.py DM_FUSE_CROSS1 [steps] [sample_images]  -> runs the training (and, if asked, sampling) leg with
VAR=0 and VAR=1, twice, alternating."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "20"
simg = sys.argv[3] if len(sys.argv) > 3 else "0"
cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "3", "--no-cpu-baseline", "--no-latent",
       "--no-elided", "--no-mnist", "--no-roofline", "--sample-images", simg, "--sample-steps", "100"]
for rnd in range(2):
    for val in ("0", "1"):
        out = subprocess.run(cmd, env=dict(os.environ, **{var: val}), capture_output=True, text=True, cwd=root)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            s = d.get("sampling") or {}
            print(f"{var}={val}: {d['value']:.0f} img/s  {d['ms_per_step']:.3f} ms/step  loss {d['final_loss']:.6f}"
                  + (f"  sampling {s['value'] * s['steps'] / 500:.2f} img/s (500-step equivalent)" if s else ""), flush=True)
        except Exception:
            print(var, val, "FAILED", out.stderr[-400:], flush=True)
