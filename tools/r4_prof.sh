#!/bin/bash
# round 4: kernel trace + stats (csv) and the gap analysis (sqlite) of the default bench line -> gpurun_out/<tag>/
tag=${1:-r04}; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra "$@" > $O/bench.log 2>&1
rocprofv3 --kernel-trace -d $O/bench_db -o t -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra "$@" > $O/bench_db.log 2>&1
python tools/trace_gaps.py $O/bench_db/t_results.db 5 grid > $O/trace_gaps.txt 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/bench_db
tail -n 1 $O/bench.log | cut -c1-400
