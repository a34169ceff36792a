#!/bin/bash
# same-box A/B of the auxiliary stream's priority inside the SAMPLING loop (forward only, UNet batch 176)
set -e
for rep in 1 2; do
  for m in 0 1; do
    echo "== DM_AUX_PRIORITY=$m sampling"; DM_AUX_PRIORITY=$m timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-latent --no-mnist --no-cpu-baseline --sample-images 88 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['sampling']['value'], d['sampling']['seconds'])"
  done
done
