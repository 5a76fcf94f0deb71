#!/bin/bash
# round 6, batch 1: (a) the long inviscid run at 128^3 against the oracle WITH the finest MAC level by colour and its passes time-skewed over slabs of 16 planes
# (the split level / slab schedule over a long run against the oracle; the switches live in the testing build), (b) the one-GPU rehearsal of configs[2]'s
# transport (eight boxes, every box-to-box copy through a 1-rank RCCL communicator) for profiles/r06_exchange_budget.md, (c) the exchange probe
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
VDN_LIB_FLAVOUR=testing VDN_MAC_SPLIT_MIN=0 VDN_MAC_SLAB=16 python tools/long_vs_oracle_inviscid.py 128 170 > $O/long_inviscid_128_split_slab_vs_oracle.txt 2>&1
tail -2 $O/long_inviscid_128_split_slab_vs_oracle.txt
VDN_FORCE_PACKED=2 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-pmc > $O/bench512_one_gpu_rccl_self.log 2>&1
grep '^{' $O/bench512_one_gpu_rccl_self.log > $O/bench512_one_gpu_rccl_self.json; tail -c 400 $O/bench512_one_gpu_rccl_self.log
python bench.py --config 512 --steps 5 --warmup 2 --skip-cpu --no-pmc > $O/bench512.log 2>&1
grep '^{' $O/bench512.log > $O/bench512.json
python tools/exchange_probe.py > $O/exchange_probe.txt 2>&1; tail -12 $O/exchange_probe.txt
