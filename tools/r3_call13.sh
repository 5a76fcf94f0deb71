#!/bin/bash
# round 3, GPU call 13: suite + poisoned suite + the judged profile artefacts + the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c13; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
bash tools/r3_poison.sh
bash tools/final_profiles_r03.sh > $O/final_profiles.log 2>&1; tail -n 16 $O/final_profiles.log | cut -c1-300
( time timeout -k 10 600 python bench.py ) > $O/bench_default.log 2>&1; tail -n 5 $O/bench_default.log | cut -c1-4000
