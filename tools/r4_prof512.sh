#!/bin/bash
# round 4: kernel statistics of 512^3 in eight 256^3 boxes on one GPU (configs[2]'s per-rank work x 8) -> gpurun_out/<tag>/
tag=${1:-r4p512}
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o a -- python3 bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/b512.log 2>&1
find $O/p -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/b512_kernel_stats.csv
rm -rf $O/p
tail -n 1 $O/b512.log | cut -c1-300
