#!/bin/bash
cd $GRAFT_REPO_ROOT; export VDN_LIB_FLAVOUR=testing VDN_PHASE_HASH=1
mkdir -p /tmp/ph
for i in 1 2 3; do timeout -k 10 100 python tools/probes/determinism_probe.py inputs-restart-regt 2 2>&1 | grep "^PHASE" > /tmp/ph/b$i.txt; done
python3 - <<'PY'
runs = [open("/tmp/ph/b%d.txt" % i).read().splitlines() for i in (1, 2, 3)]
a = runs[0]
for i in (1, 2):
    b = runs[i]
    diff = [k for k in range(min(len(a), len(b))) if a[k] != b[k]]
    print("run %d: %d of %d lines differ; the first twelve:" % (i + 1, len(diff), len(a)))
    for k in diff[:12]: print("   ", b[k][:150], " | run 1:", a[k].split()[-3])
PY
