#!/bin/bash
# round 3, GPU call 39: two-step damping of the nodal pre-smoothing sweeps: full suite, bench A/B, 512, amr2, amr3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c39; mkdir -p $O
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -n 15 $O/pytest.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do echo "== hg_pre_pair $v"; timeout -k 10 300 python bench.py --steps 10 --warmup 2 --skip-cpu --no-extra --hg-pre-pair $v 2>&1 | tail -n 1 | cut -c1-760 || exit 1; done > $O/bench_ab.log 2>&1 && cat $O/bench_ab.log &&
for c in 512 amr2 amr3; do echo "== $c"; timeout -k 10 400 python bench.py --config $c --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-900 || exit 1; done > $O/bench_other.log 2>&1 && cat $O/bench_other.log
