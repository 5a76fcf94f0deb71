#!/bin/bash
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04gaps6; mkdir -p $O
rocprofv3 --kernel-trace -d $O/db -o t -- python3 bench.py --config amr2 --steps 6 --warmup 1 --skip-cpu --no-extra > $O/amr2.log 2>&1
for s in 1 2 3 4 5; do python tools/trace_gaps.py $O/db/t_results.db $s | head -1; done
python tools/trace_gaps.py $O/db/t_results.db 5 > $O/amr2_step5.txt
rm -rf $O/db
