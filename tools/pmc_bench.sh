#!/bin/bash
# separate rocprofv3 --pmc passes (never combined with other trace domains) over a short bench run; csv per pass
# usage: pmc_bench.sh tag "COUNTERS1" "COUNTERS2" ...     (env for the bench is inherited)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --skip-cpu > $GRAFT_REPO_ROOT/gpurun_out/$tag.p$i.log 2>&1
  echo "pass $i ($ctr) rc=$?"
done
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/$tag > gpurun_out/$tag.txt 2>&1; head -c 6000 gpurun_out/$tag.txt
