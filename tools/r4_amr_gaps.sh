#!/bin/bash
# round 4: where the GPU idles in a step of the tagged hierarchies (kernel trace -> gap analysis) -> gpurun_out/<tag>/
tag=${1:-r04amrgaps}; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
for c in amr2 amr3; do
  rocprofv3 --kernel-trace -d $O/db_$c -o t -- python3 bench.py --config $c --steps 3 --warmup 1 --skip-cpu --no-extra --no-pmc "$@" > $O/$c.log 2>&1
  python tools/trace_gaps.py $O/db_$c/t_results.db 2 grid > $O/${c}_trace_gaps.txt 2>&1
  rm -rf $O/db_$c
  head -n 2 $O/${c}_trace_gaps.txt | cut -c1-600
done
