#!/bin/bash
# same-box A/B of the GroupNorm forward variants: two passes (0), whole-CU one pass (1), half-CU one pass (2)
set -e
for m in 0 1 2; do
  echo "== DM_GN_ONEPASS=$m hbmbench"; DM_GN_ONEPASS=$m timeout -k 10 120 python tools/hbmbench.py 2>/dev/null | grep gn_silu_fwd
done
for rep in 1 2; do
  for m in 0 2 1; do
    echo "== DM_GN_ONEPASS=$m train"; DM_GN_ONEPASS=$m timeout -k 10 200 python bench.py --steps 20 --no-latent --no-mnist --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
