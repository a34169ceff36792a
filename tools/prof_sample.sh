#!/bin/bash
# rocprofv3 kernel trace of the sampling leg (88 images = UNet batch 176, a short 40-step schedule), summarised per kernel
set -e
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_sample_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 \
  --no-latent --no-mnist --no-cpu-baseline --sample-images 88 --sample-steps 40 --no-elided --no-fp32 --no-roofline > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
python tools/summarize_rocpd.py $out 50 > gpurun_out/${tag}_sample_kernel_stats.csv
head -45 gpurun_out/${tag}_sample_kernel_stats.csv | cut -c1-190
