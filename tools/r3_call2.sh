#!/bin/bash
# round 3, GPU call 2: multirank tests (rebuilt test double), non-temporal probes (cc colour pass, nodal march), per-step exchange counts
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_multirank_gpu.py tests/test_fortran_gpu.py tests/test_inputs_gpu.py tests/test_plotfile_gpu.py tests/test_projection_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
for nt in 0 1; do for n in 256 128; do echo -n "cc nt=$nt n=$n: "; VDN_CC_NT=$nt timeout -k 10 120 python tools/smoother_probe.py $n 200 | head -n 1; done; done > $O/cc_nt.log 2>&1; cat $O/cc_nt.log
for var in 0 1 2 3; do for n in 256 128; do echo -n "nd var=$var n=$n: "; VDN_ND_VAR=$var timeout -k 10 120 python tools/nd_probe.py $n 2>&1 | grep VDN_ND_BENCH; done; done > $O/nd_var.log 2>&1; cat $O/nd_var.log
for nt in 0 1; do VDN_CC_NT=$nt VDN_ND_VAR=$nt timeout -k 10 300 python bench.py --steps 5 --warmup 2 --skip-cpu > $O/bench_nt$nt.log 2>&1; tail -n 1 $O/bench_nt$nt.log | cut -c1-700; done
VDN_FORCE_PACKED=2 timeout -k 10 400 python bench.py --config 512 --steps 2 --warmup 1 --skip-cpu > $O/bench512_packed2.log 2>&1; tail -n 1 $O/bench512_packed2.log | cut -c1-2500
VDN_FORCE_PACKED=2 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --skip-cpu > $O/bench256_packed2.log 2>&1; tail -n 1 $O/bench256_packed2.log | cut -c1-2500
