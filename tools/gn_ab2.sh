#!/bin/bash
# same-box A/B of the GroupNorm forward variants inside the SAMPLING loop (forward only, UNet batch 176)
set -e
for rep in 1 2; do
  for m in 0 2; do
    echo "== DM_GN_ONEPASS=$m sampling"; DM_GN_ONEPASS=$m timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-latent --no-mnist --no-cpu-baseline --sample-images 88 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['sampling']['value'], d['sampling']['seconds'])"
  done
done
