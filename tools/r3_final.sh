#!/bin/bash
# round 3, the judged artefacts: default bench line, kernel statistics / gap analysis / PMC passes of the same command, the smoother probe,
# exchange counts of the 512^3 configuration through the RCCL self-communicator, kernel statistics of the two hierarchies -> gpurun_out/r03/
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_default.log 2>&1 || { tail -n 5 $O/bench_default.log; exit 1; }
tail -n 1 $O/bench_default.log > $O/bench_default_line.json; cut -c1-400 $O/bench_default_line.json
bash tools/final_profiles_r03.sh > $O/final_profiles.log 2>&1 || { tail -n 5 $O/final_profiles.log; exit 1; }
tail -n 3 $O/final_profiles.log | cut -c1-300
VDN_FORCE_PACKED=2 timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-extra > $O/bench512_rccl_self.log 2>&1 || { tail -n 5 $O/bench512_rccl_self.log; exit 1; }
tail -n 1 $O/bench512_rccl_self.log > $O/bench512_one_gpu_rccl_self.json; cut -c1-300 $O/bench512_one_gpu_rccl_self.json
for c in amr2 amr3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -o a -- python3 bench.py --config $c --steps 3 --warmup 1 --skip-cpu --no-extra > $O/$c.log 2>&1 || { tail -n 5 $O/$c.log; exit 1; }
  tail -n 1 $O/$c.log | cut -c1-200
done
