#!/bin/bash
# round 3, GPU call 19: nested-iteration start of the cell-centred solve (mac_fmg): full suite, bench A/B, amr2, 512
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c19; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -n 25 $O/pytest.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do echo "== mac_fmg $v"; timeout -k 10 300 python bench.py --steps 10 --warmup 2 --skip-cpu --no-extra --mac-fmg $v 2>&1 | tail -n 1 | cut -c1-760 || exit 1; done > $O/bench_ab.log 2>&1 && cat $O/bench_ab.log &&
for v in 0 1; do echo "== amr2 mac_fmg $v"; timeout -k 10 300 python bench.py --config amr2 --steps 5 --warmup 2 --skip-cpu --no-extra --mac-fmg $v 2>&1 | tail -n 1 | cut -c1-900 || exit 1; done > $O/amr2_ab.log 2>&1 && cat $O/amr2_ab.log &&
timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-800
