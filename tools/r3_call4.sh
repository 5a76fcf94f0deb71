#!/bin/bash
# round 3, GPU call 4: full GPU suite after hg_nub 8 / force reuse / temp descriptors; default bench line with extra workloads (wall time)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
( time timeout -k 10 600 python bench.py ) > $O/bench_default.log 2>&1; tail -n 5 $O/bench_default.log | cut -c1-3000
VDN_NO_FORCE_REUSE=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640
