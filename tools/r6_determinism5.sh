#!/bin/bash
# reproducibility at the bench's size: inputs-restart-regt with a 256^3 base, max_grid_size 64, three levels, viscous, regridding; four steps, three processes
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import re
t = open("tests/golden/inputs/inputs-restart-regt").read()
for a in "xyz": t = re.sub(r"n_cell%s\s*=\s*\d+" % a, "n_cell%s = 256" % a, t)
t = re.sub(r"max_grid_size\s*=\s*\d+", "max_grid_size = 64", t)
open("tests/golden/inputs/_det256", "w").write(t)
PY
rm -f /tmp/det.txt
for i in 1 2 3; do timeout -k 10 300 python tools/probes/determinism_probe.py _det256 4 2>/dev/null | grep "^_det" | cut -c1-300 >> /tmp/det.txt; done
echo "[256^3 base] runs $(wc -l < /tmp/det.txt), distinct $(sort -u /tmp/det.txt | wc -l)"; sort -u /tmp/det.txt | cut -c1-160
rm -f tests/golden/inputs/_det256
