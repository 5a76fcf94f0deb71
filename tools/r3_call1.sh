#!/bin/bash
# round 3, GPU call 1: full GPU suite, nodal-march launch-shape probes, bench, exchange probe + per-step exchange counts
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c1; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
for n in 256 128; do
for rem in 0 1; do for kc in 0 12 16 20 26 32; do
  echo -n "n=$n rem=$rem kc=$kc: "; VDN_ND_REM=$rem VDN_ND_KC=$kc timeout -k 10 120 python tools/nd_probe.py $n 2>&1 | grep VDN_ND_BENCH
done; done; done > $O/nd_probe.log 2>&1
cat $O/nd_probe.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu > $O/bench.log 2>&1; tail -n 1 $O/bench.log | cut -c1-1500
timeout -k 10 600 python tools/exchange_probe.py > $O/exchange_probe.log 2>&1; cat $O/exchange_probe.log
VDN_FORCE_PACKED=2 timeout -k 10 400 python bench.py --config 512 --steps 2 --warmup 1 --skip-cpu > $O/bench512_packed2.log 2>&1; tail -n 1 $O/bench512_packed2.log | cut -c1-2500
