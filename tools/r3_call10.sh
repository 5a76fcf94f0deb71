#!/bin/bash
# round 3, GPU call 10: update with its forcing formed in place, paired nodal restriction: full suite, bench A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c10; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for v in "VDN_NO_FORCE_REUSE=1 VDN_ND_RESTRICT_PAIR=0" "VDN_ND_RESTRICT_PAIR=0" "VDN_ND_RESTRICT_PAIR=1"; do echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640; done > $O/bench_ab.log 2>&1; cat $O/bench_ab.log
