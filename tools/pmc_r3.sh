#!/bin/bash
# Round-3 PMC passes (separate runs per counter block, as MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 of the 4 TCC
# slots, WRITE_SIZE 2): HBM traffic of the dominant conv kernel and of the default GroupNorm forward kernels.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r3
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $O/conv_$c -- python3 $R/tools/pmc_conv.py > $O/conv_$c.out 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $O/gn_$c -- python3 $R/tools/pmc_hbm2.py > $O/gn_$c.out 2>&1
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== conv $c"; python tools/pmcsum.py $O/conv_$c $c | head -4
  echo "== gn/cross1 $c"; python tools/pmcsum.py $O/gn_$c $c | head -8
done
grep algorithmic $O/conv_FETCH_SIZE.out $O/gn_FETCH_SIZE.out
