#!/bin/bash
cd $GRAFT_REPO_ROOT; export VDN_LIB_FLAVOUR=testing VDN_PHASE_HASH=1
N=${1:-6}; mkdir -p /tmp/ph
for i in $(seq $N); do timeout -k 10 200 python tools/probes/determinism_probe.py ${PROBE_ARGS:-} 2>&1 | grep "^PHASE" > /tmp/ph/$i.txt; done
python3 - $N <<'PY'
import sys, collections
N = int(sys.argv[1])
runs = [open("/tmp/ph/%d.txt" % i).read().splitlines() for i in range(1, N + 1)]
a = runs[0]
for i in range(1, N):
    b = runs[i]
    diff = [k for k in range(min(len(a), len(b))) if a[k] != b[k]]
    if not diff: print("run %d identical" % (i + 1)); continue
    d0 = diff[0]
    call = sum(1 for ln in a[:d0 + 1] if "uold at entry" in ln)
    print("run %d: %d lines differ; first in advance call %d:" % (i + 1, len(diff), call))
    for k in diff[:10]:
        print("    " + b[k][:135] + " | run 1 " + (a[k].split()[-3] if a[k].startswith("PHASEBOX") else a[k].split()[-1]))
PY
