#!/bin/bash
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default_line.json; tail -c 200 $O/bench_default.log; echo
python bench.py --config 512 --steps 5 --warmup 2 --skip-cpu --no-pmc > $O/bench512.log 2>&1; grep '^{' $O/bench512.log > $O/bench512.json
VDN_FORCE_PACKED=2 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-pmc --no-extra > $O/bench512_one_gpu_rccl_self.log 2>&1; grep '^{' $O/bench512_one_gpu_rccl_self.log > $O/bench512_one_gpu_rccl_self.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b512 -o b -- python3 bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-pmc --no-extra > $O/b512.log 2>&1
cp $(find $O/b512 -name "*kernel_stats.csv" | head -1) $O/r06_bench512_kernel_stats.csv
python - <<'PY'
import json
for f in ("gpurun_out/r06/bench_default_line.json","gpurun_out/r06/bench512.json","gpurun_out/r06/bench512_one_gpu_rccl_self.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]
    print(f, d["ms_per_step"], c["phase_ms_per_step"], (c.get("one_box_512") or {}).get("ms_per_step"), [e["ms_per_step"] for e in d["extra_workloads"]])
PY
