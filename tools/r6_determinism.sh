#!/bin/bash
# process-to-process reproducibility of inputs-restart-regt under the testing switches: N runs per configuration, the distinct hash sequences counted
cd $GRAFT_REPO_ROOT; export VDN_LIB_FLAVOUR=testing
N=${1:-6}
for cfg in "" "VDN_KEEP_SETS=0" "VDN_NO_GRAPHS=1" "VDN_KEEP_OFF=8" "VDN_KEEP_OFF=4" "VDN_KEEP_OFF=1" "VDN_KEEP_OFF=2" "VDN_MLCC_RHO=0" "VDN_MLCC_FUSE1=0"; do
  rm -f /tmp/det.txt
  for i in $(seq $N); do env $cfg timeout -k 10 100 python tools/probes/determinism_probe.py 2>&1 | grep "^inputs" >> /tmp/det.txt; done
  echo "[$cfg] runs $(wc -l < /tmp/det.txt), distinct $(sort -u /tmp/det.txt | wc -l)"
done
