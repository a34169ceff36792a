#!/bin/bash
# same-box A/B of the GroupNorm BACKWARD variants inside the training step (DM_GN_ONEPASS_BWD = largest slab, in 16-byte
# pieces per thread, the one-pass backward takes; 0 = two passes everywhere)
set -e
for rep in 1 2 3 4 5; do
  for m in 0 5 9 18; do
    echo -n "DM_GN_ONEPASS_BWD=$m  "; DM_GN_ONEPASS_BWD=$m timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-latent --no-mnist --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
