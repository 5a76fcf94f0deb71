#!/bin/bash
# round 3, GPU call 8: full suite (new full-size hierarchy tests, boussinesq), the poisoned-arena run, a kernel profile of the tagged two-level hierarchy
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c8; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 16 $O/pytest.log | cut -c1-200
bash tools/r3_poison.sh
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_amr2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config amr2 --steps 3 --warmup 1 --skip-cpu > $GRAFT_REPO_ROOT/$O/prof_amr2.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof_amr2 -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/amr2_kernel_stats.csv; head -n 40 $O/amr2_kernel_stats.csv | cut -c1-170; tail -n 1 $O/prof_amr2.log | cut -c1-900
