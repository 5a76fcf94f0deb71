#!/bin/bash
# round 3: the AMR tests and the full-size property tests, the two- and three-level bench lines, kernel statistics of both -> gpurun_out/r3_amr/
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_amr; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_amr_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -n 12 $O/pytest.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for c in amr2 amr3; do echo "== $c"; timeout -k 10 400 python bench.py --config $c --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-1000 || exit 1; done > $O/bench.log 2>&1 && cat $O/bench.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o a -- python3 bench.py --config amr2 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/prof2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -o a -- python3 bench.py --config amr3 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/prof3.log 2>&1 && tail -n 1 $O/prof3.log | cut -c1-200
