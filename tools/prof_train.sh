#!/bin/bash
# rocprofv3 kernel trace of the training leg (same command as the bench's headline leg, secondary legs off), summarised to
# a per-kernel CSV; the bench JSON of the profiled run rides along.  Usage: bash tools/prof_train.sh <tag>
set -e
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 \
  --no-latent --no-mnist --no-cpu-baseline --sample-images 0 --no-elided --no-fp32 > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
python tools/summarize_rocpd.py $out 70 > gpurun_out/${tag}_train_kernel_stats.csv
cp $out/bench.json gpurun_out/${tag}_train_bench_profiled.json
head -30 gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-170
