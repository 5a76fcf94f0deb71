#!/bin/bash
# round 4: where the GPU idles in a step of 512^3 in eight 256^3 boxes (kernel trace -> gap analysis) -> gpurun_out/<tag>/
tag=${1:-r04gaps512}; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
rocprofv3 --kernel-trace -d $O/db -o t -- python3 bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-extra "$@" > $O/b512.log 2>&1
python tools/trace_gaps.py $O/db/t_results.db 2 grid > $O/b512_trace_gaps.txt 2>&1
rm -rf $O/db
head -n 1 $O/b512_trace_gaps.txt
