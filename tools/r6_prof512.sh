#!/bin/bash
# kernel stats of the 512^3-in-eight-boxes workload (configs[2] on one GPU)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b512 -o b -- python3 bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-pmc > $O/b512.log 2>&1
ls $O/b512/*/ 2>/dev/null | head
f=$(find $O/b512 -name "*kernel_stats.csv" | head -1); cp $f $O/r06_bench512_kernel_stats.csv; head -50 $f | cut -c1-200
