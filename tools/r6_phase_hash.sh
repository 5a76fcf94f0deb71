#!/bin/bash
# N runs of inputs-restart-regt with a checksum at every phase boundary of every step (VDN_PHASE_HASH=1): the first line at which a run leaves run 1
cd $GRAFT_REPO_ROOT; export VDN_LIB_FLAVOUR=testing VDN_PHASE_HASH=1
N=${1:-6}; mkdir -p /tmp/ph
for i in $(seq $N); do timeout -k 10 100 python tools/probes/determinism_probe.py 2>&1 | grep "^PHASE" > /tmp/ph/$i.txt; done
python3 - $N <<'PY'
import sys
N = int(sys.argv[1])
runs = [open("/tmp/ph/%d.txt" % i).read().splitlines() for i in range(1, N + 1)]
print("lines per run:", [len(r) for r in runs])
per_step = None
for i in range(1, N):
    a, b = runs[0], runs[i]
    d = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), None)
    if d is None:
        print("run %d: identical to run 1" % (i + 1)); continue
    # which advance call is it?  count the "uold at entry" lines before
    call = sum(1 for ln in a[:d + 1] if "uold at entry" in ln)
    print("run %d: first difference at line %d, advance call %d: %s   | run 1: %s" % (i + 1, d, call, b[d][:140], a[d].split()[-3] if a[d].startswith("PHASEBOX") else a[d].split()[-1]))
    prev = [x for x in a[:d] if x.startswith("PHASE ")][-3:]
    print("      the phases before: " + " | ".join(x.replace("PHASE ", "").strip()[:24] for x in prev))
PY
