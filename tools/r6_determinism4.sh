#!/bin/bash
# the same reproducibility check on larger hierarchies: inputs-restart-regt with a 128^3 base (three levels, max_grid_size 32: some 700 boxes), the original, and inputs_RayleighTaylor_2d (four levels, periodic, extruded)
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import re
t = open("tests/golden/inputs/inputs-restart-regt").read()
for a in "xyz": t = re.sub(r"n_cell%s\s*=\s*\d+" % a, "n_cell%s = 128" % a, t)
open("tests/golden/inputs/_det128", "w").write(t)
PY
for name in "_det128 5" "inputs-restart-regt 8" "inputs_RayleighTaylor_2d 12"; do
  rm -f /tmp/det.txt
  for i in $(seq ${1:-8}); do timeout -k 10 200 python tools/probes/determinism_probe.py $name 2>/dev/null | grep "^_det\|^inputs" | cut -c1-300 >> /tmp/det.txt; done
  echo "[$name] runs $(wc -l < /tmp/det.txt), distinct $(sort -u /tmp/det.txt | wc -l)"
done
rm -f tests/golden/inputs/_det128
