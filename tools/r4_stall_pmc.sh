#!/bin/bash
# round 4: where the waves of the weakest kernels wait -- separate rocprofv3 --pmc passes (kernel trace only) over a short bench run, summarised per kernel
# by tools/pmc_summary.py -> gpurun_out/<tag>.txt.  Counters: wave / busy / wait cycles, active cycles per instruction class, instruction counts, LDS conflicts
tag=${1:-r04stall}
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "MemUnitStalled MeanOccupancyPerCU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR TCP_PENDING_STALL_CYCLES_sum SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --skip-cpu --no-extra --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/$tag.p$i.log 2>&1
  echo "pass $i ($ctr) rc=$?"
done
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/$tag > gpurun_out/$tag.txt 2>&1; head -c 3000 gpurun_out/$tag.txt
