#!/bin/bash
# round 3, GPU call 12: update inside the mkflux march: full suite, bench A/B, kernel durations
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c12; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for v in "VDN_GOD_UPDATE=0" "VDN_GOD_UPDATE=1" "VDN_GOD_UPDATE=0" "VDN_GOD_UPDATE=1"; do echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640; done > $O/bench_ab.log 2>&1; cat $O/bench_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $O/bench.log 2>&1
f=$(find $O/bench -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/kernel_stats.csv; grep -n "mk_F_m\|kk_update\|update_vf" $O/kernel_stats.csv | cut -c1-180
