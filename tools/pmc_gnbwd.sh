#!/bin/bash
# PMC passes (one counter block per run) for the GroupNorm backward: one-pass kernel and the two-pass pair
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gnbwd
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $O/one_$c -- python3 $R/tools/pmc_gnbwd.py > $O/one_$c.out 2>&1
done
export DM_GN_ONEPASS_BWD=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $O/two_$c -- python3 $R/tools/pmc_gnbwd.py > $O/two_$c.out 2>&1
done
cd $R
for v in one two; do for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $v $c"; python tools/pmcsum.py $O/${v}_$c $c | grep -i "gn_" | head -6
done; done
grep -h algorithmic $O/one_FETCH_SIZE.out $O/two_FETCH_SIZE.out
