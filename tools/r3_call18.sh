#!/bin/bash
# round 3, GPU call 18: planes per workgroup of the box-batched launches (VDN_BATCH_PPW) on the two-level bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c18; mkdir -p $O
for v in 1 2 4 8 16; do echo "== VDN_BATCH_PPW $v"; VDN_BATCH_PPW=$v timeout -k 10 300 python bench.py --config amr2 --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-900 || exit 1; done > $O/ppw.log 2>&1
grep -o "VDN_BATCH_PPW [0-9]*\|\"ms_per_step\": [0-9.]*\|\"phase_ms_per_step\": {[^}]*}" $O/ppw.log
export VDN_BATCH_PPW=8
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o a -- python3 bench.py --config amr2 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/prof.log 2>&1 && tail -n 1 $O/prof.log | cut -c1-200
