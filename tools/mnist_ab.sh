#!/bin/bash
# MNIST leg (eager / micro-step graph / whole-step graph) under developer switches, same box
set -e
run() { timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-latent --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['mnist']; print(d['eager']['ms_per_step'], d['hip_graph']['ms_per_step'], d['hip_graph']['host_issue_ms_per_step'], d['hip_graph_whole_step']['ms_per_step'], d['hip_graph_whole_step']['host_issue_ms_per_step'])"; }
for rep in 1 2; do
  echo "== default"; run
  echo "== aux prio 0"; DM_AUX_PRIORITY=0 run
  echo "== gn bwd two-pass"; DM_GN_ONEPASS_BWD=0 run
  echo "== both off"; DM_AUX_PRIORITY=0 DM_GN_ONEPASS_BWD=0 run
done
