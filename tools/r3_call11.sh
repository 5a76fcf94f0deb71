#!/bin/bash
# round 3, GPU call 11: polling read-back A/B, restriction kernels in a profile
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c11; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_projection_gpu.py tests/test_advance_gpu.py tests/test_multirank_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
for v in "VDN_POLL=0" "VDN_POLL=1" "VDN_POLL=0" "VDN_POLL=1"; do echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640; done > $O/bench_ab.log 2>&1; cat $O/bench_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $O/bench.log 2>&1
f=$(find $O/bench -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/kernel_stats.csv; grep -n "restrict\|prolong" $O/kernel_stats.csv | cut -c1-160
VDN_ND_RESTRICT_PAIR=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench0 -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $O/bench0.log 2>&1
f=$(find $O/bench0 -name "*kernel_stats.csv" | head -n 1); grep -n "restrict" $f | cut -c1-160
