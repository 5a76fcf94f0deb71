#!/bin/bash
# round 3, GPU call 17: composite nodal solve, first coarse correction from the nested iteration: AMR tests, bench amr2 A/B, kernel stats
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c17; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_amr_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -n 12 $O/pytest.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for v in 0 1; do echo "== hg_fmg $v"; timeout -k 10 300 python bench.py --config amr2 --steps 5 --warmup 2 --skip-cpu --no-extra --hg-fmg $v 2>&1 | tail -n 1 | cut -c1-1200 || exit 1; done > $O/bench_ab.log 2>&1 && cat $O/bench_ab.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o a -- python3 bench.py --config amr2 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/prof.log 2>&1 && tail -n 1 $O/prof.log | cut -c1-300
