#!/bin/bash
# where does the run-to-run difference of inputs-restart-regt start, and which ingredient does it need?  (first differing step over N runs per variant)
cd $GRAFT_REPO_ROOT
N=${1:-6}
mk() { python3 - "$@" <<'PY'
import re, sys
t = open("tests/golden/inputs/inputs-restart-regt").read()
for a in sys.argv[1:]:
    k, v = a.split("=")
    t = re.sub(r"%s\s*=\s*[-\w.]+" % k, "%s = %s" % (k, v), t)
open("tests/golden/inputs/_det_variant", "w").write(t)
PY
}
for var in "regrid_int=2" "regrid_int=-1" "visc_coef=0.0" "max_levs=2" "max_levs=2 visc_coef=0.0" "max_grid_size=64" "n_cellx=32 n_celly=32 n_cellz=32"; do
  mk $var
  rm -f /tmp/det.txt
  for i in $(seq $N); do timeout -k 10 100 python tools/probes/determinism_probe.py _det_variant 8 2>&1 | grep "^_det" >> /tmp/det.txt; done
  python3 - "$var" <<'PY'
import sys
rows = [ln.split()[1:] for ln in open("/tmp/det.txt")]
first = None
for s in range(len(rows[0])):
    if len(set(r[s] for r in rows)) > 1:
        first = s + 1; break
print("[%s] runs %d, distinct sequences %d, first differing step %s" % (sys.argv[1], len(rows), len(set(tuple(r) for r in rows)), first))
PY
done
rm -f tests/golden/inputs/_det_variant
