#!/bin/bash
# round 4: kernel statistics of the tagged two- and three-level hierarchies -> gpurun_out/<tag>/
tag=${1:-r04amr}
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
for c in amr2 amr3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -o a -- python3 bench.py --config $c --steps 3 --warmup 1 --skip-cpu --no-extra --no-pmc > $O/$c.log 2>&1
  find $O/$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${c}_kernel_stats.csv
  rm -rf $O/$c
  tail -n 1 $O/$c.log | cut -c1-300
done
