#!/bin/bash
# rocprofv3 kernel trace of the fp32-compute-mode training leg (3 steps), summarised per kernel
set -e
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_fp32_$tag
rm -rf $out && mkdir -p $out
DM_COMPUTE_DTYPE=fp32 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 \
  --no-latent --no-mnist --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 --no-roofline > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
python tools/summarize_rocpd.py $out 40 > gpurun_out/${tag}_fp32_kernel_stats.csv
head -30 gpurun_out/${tag}_fp32_kernel_stats.csv | cut -c1-180
cat $out/bench.json | cut -c1-400
