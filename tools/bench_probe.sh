#!/bin/bash
# one GPU call: parity tests of the solvers, bench with / without graphs, kernel stats, smoother probes
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_projection_gpu.py tests/test_advance_gpu.py tests/test_multibox_gpu.py -x -q > $O/r2_t2.log 2>&1; echo "pytest rc=$?" >> $O/r2_t2.log
python bench.py --steps 5 --warmup 2 --skip-cpu > $O/r2_b_graph.log 2>&1
VDN_NO_GRAPHS=1 python bench.py --steps 5 --warmup 2 --skip-cpu > $O/r2_b_nograph.log 2>&1
for n in 256 128 64; do python tools/smoother_probe.py $n 200 stored | head -1; python tools/smoother_probe.py $n 200 | head -1; done > $O/r2_probe.log 2>&1
python - <<'PY' > $O/r2_arena.log 2>&1
import sys, ctypes as C
sys.path.insert(0, ".")
from varden_amd import capi, driver
G = driver.Varden(256, [[15, 15]] * 3, capi.default_params(cflfac=0.9), init_shrink=0.1, init_iter=1)
G.step(); G.step()
a, b = C.c_size_t(), C.c_size_t()
capi.load().vdn_arena_stats(C.byref(a), C.byref(b))
print("arena reserved %.2f GB peak %.2f GB; per-field (262^3*8) %.3f GB => peak = %.1f fields" % (a.value / 1e9, b.value / 1e9, 262 ** 3 * 8 / 1e9, b.value / (264 ** 3 * 8)))
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r2_prof1 -o r2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --skip-cpu > $GRAFT_REPO_ROOT/$O/r2_prof1.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O/r2_prof1 | head -20
tail -3 $O/r2_t2.log; cat $O/r2_b_graph.log | tail -1; cat $O/r2_b_nograph.log | tail -1; cat $O/r2_probe.log; cat $O/r2_arena.log
