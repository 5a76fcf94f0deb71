#!/bin/bash
# rocprofv3 kernel trace of the default bench line -> gpurun_out/<tag>/ (sqlite) ; usage: prof_bench.sh tag [bench args]
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --skip-cpu "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/trace_gaps.py gpurun_out/$tag/t_results.db 5 grid > gpurun_out/$tag.txt 2>&1
tail -n 1 gpurun_out/$tag.log | cut -c1-300
