#!/bin/bash
# round 3, GPU call 16: nested-iteration start of the nodal solve: full suite, bench A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c16; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 12 $O/pytest.log | cut -c1-220
for v in 0 1 0 1; do echo "== hg_fmg $v"; timeout -k 10 300 python bench.py --steps 10 --warmup 2 --skip-cpu --no-extra --hg-fmg $v 2>&1 | tail -n 1 | cut -c1-760; done > $O/bench_ab.log 2>&1; cat $O/bench_ab.log
timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu 2>&1 | tail -n 1 | cut -c1-700
