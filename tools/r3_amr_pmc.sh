#!/bin/bash
# round 3: HBM counters of the two-level bench line (separate passes, kernel trace only) -> gpurun_out/r3_amr_pmc/
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_amr_pmc; mkdir -p $O
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/p$i -o p -- python3 bench.py --config amr2 --steps 2 --warmup 1 --skip-cpu --no-extra > $O/pmc$i.log 2>&1 || { tail -n 5 $O/pmc$i.log; exit 1; }
done
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
head -n 30 $O/pmc_summary.txt | cut -c1-220
