#!/bin/bash
# every kernel of the library with its VGPR count and scratch bytes per lane (hipcc -Rpass-analysis=kernel-resource-usage); kernels that use
# scratch are listed at the end -- a descriptor copied by value and indexed dynamically ends up there (see desc_in_constant in vdn_dev.h)
cd "$(dirname "$0")/../varden_amd/csrc" || exit 1
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/chk_$$.o 2>&1 \
    | awk -v F=$f '/Function Name:/ { n=$(NF-1) } / VGPRs:/ { v=$(NF-1) } /ScratchSize/ { print F, n, "vgprs", v, "scratch", $(NF-1) }'
done | awk '{ print } $6 > 0 { bad[++nb] = $0 } END { print "---- kernels with scratch:"; for (i = 1; i <= nb; i++) print bad[i] }'
rm -f /tmp/chk_$$.o
