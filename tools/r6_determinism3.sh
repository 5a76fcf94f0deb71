#!/bin/bash
cd $GRAFT_REPO_ROOT
N=${1:-8}
for cfg in "AMD_OPT_FLUSH=0" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=1"; do
rm -f /tmp/det.txt
for i in $(seq $N); do env $cfg timeout -k 10 100 python tools/probes/determinism_probe.py 2>/dev/null | grep "^inputs" | cut -c1-260 >> /tmp/det.txt; done
echo "[$cfg] runs $(wc -l < /tmp/det.txt), distinct $(sort -u /tmp/det.txt | wc -l)"
done
