#!/bin/bash
# round 3, GPU call 5: LDS-tiled small-level kernels of the nodal multigrid: parity, bits, bench, kernel durations
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c5; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_projection_gpu.py tests/test_advance_gpu.py tests/test_amr_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for v in "VDN_MG_LDS=0" "VDN_MG_LDS=1" "VDN_MG_LDS_MAX=32"; do echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu --no-extra 2>&1 | tail -n 1 | cut -c1-640; done > $O/bench_lds.log 2>&1; cat $O/bench_lds.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --skip-cpu --no-extra > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/kernel_stats.csv; head -n 45 $O/kernel_stats.csv | cut -c1-160
