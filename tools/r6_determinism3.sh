#!/bin/bash
cd $GRAFT_REPO_ROOT
N=${1:-10}
rm -f /tmp/det.txt
for i in $(seq $N); do timeout -k 10 100 python tools/probes/determinism_probe.py 2>/dev/null | grep "^inputs" | cut -c1-260 >> /tmp/det.txt; done
echo "runs $(wc -l < /tmp/det.txt), distinct $(sort -u /tmp/det.txt | wc -l)"; sort -u /tmp/det.txt | cut -c1-200
