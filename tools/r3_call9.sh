#!/bin/bash
# round 3, GPU call 9: the fixed full-size tests; kernel statistics + gap analysis of the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c9; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $O/bench.log 2>&1
rocprofv3 --kernel-trace -d $O/bench_db -o t -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $O/bench_db.log 2>&1
python tools/trace_gaps.py $O/bench_db/t_results.db 5 grid > $O/trace_gaps.txt 2>&1
f=$(find $O/bench -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/kernel_stats.csv
head -n 70 $O/trace_gaps.txt | cut -c1-170
