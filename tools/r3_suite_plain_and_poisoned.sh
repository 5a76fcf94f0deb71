#!/bin/bash
# round 3, the whole GPU suite, plain and with the arena poisoned, after the AMR work of calls 17-28
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_suite; mkdir -p $O
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -n 6 $O/pytest.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
VDN_ARENA_POISON=1 timeout -k 10 560 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest_poison.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_poison.log; tail -n 6 $O/pytest_poison.log | cut -c1-250
exit $rc
