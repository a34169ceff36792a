#!/bin/bash
# same-box A/B of stream priorities for the auxiliary (dead-attention) and weight-gradient streams inside the training step
set -e
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-latent --no-mnist --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== default"; run
  echo "== aux low"; DM_AUX_PRIORITY=1 run
  echo "== aux low, side high"; DM_AUX_PRIORITY=1 DM_SIDE_PRIORITY=-1 run
  echo "== aux high"; DM_AUX_PRIORITY=-1 run
  echo "== side low"; DM_SIDE_PRIORITY=1 run
done
