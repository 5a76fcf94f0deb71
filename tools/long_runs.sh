#!/bin/bash
# long runs (round 4: after the solver changes; round 5: after the flux restriction of mkflux.f90:137-146 and the one-barrier marches): HIP against the oracle over
# 150 steps on one level and 80 on two, and the reference's 3-D inputs files for a few dozen steps each
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/long; mkdir -p $O
timeout -k 10 400 python tools/long_vs_oracle.py 150 > $O/long1.log 2>&1 || { tail -n 5 $O/long1.log; exit 1; }; tail -n 6 $O/long1.log
timeout -k 10 400 python tools/long_vs_oracle_amr.py 80 > $O/long2.log 2>&1 || { tail -n 5 $O/long2.log; exit 1; }; tail -n 4 $O/long2.log
for f in inputs_bubble_3d:60 inputs_3d-regt:40 inputs_RayleighTaylor_3d:30 inputs_advect_3d:30 inputs_vortextube_3d:20; do
  n=${f%%:*}; s=${f##*:}
  timeout -k 10 300 python tools/run_inputs.py tests/golden/inputs/$n $s $O/run_$n > $O/$n.log 2>&1 || { tail -n 8 $O/$n.log; exit 1; }
  echo "== $n"; tail -n 4 $O/$n.log | cut -c1-200
  rm -rf $O/run_$n
done
