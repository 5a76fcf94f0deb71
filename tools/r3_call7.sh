#!/bin/bash
# round 3, GPU call 7: projection fast paths: full suite, the suite again with the arena poisoned, A/B bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c7; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
VDN_ARENA_POISON=1 timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest_poison.log 2>&1; echo "pytest rc=$?" >> $O/pytest_poison.log; tail -n 45 $O/pytest_poison.log | cut -c1-200
for v in "VDN_HG_FAST=0 VDN_MAC_FAST=0 VDN_ND_LEAN=0" "VDN_HG_FAST=1 VDN_MAC_FAST=0" "VDN_HG_FAST=1 VDN_MAC_FAST=1"; do echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640; done > $O/bench_ab.log 2>&1; cat $O/bench_ab.log
