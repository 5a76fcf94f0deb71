#!/bin/bash
# the round's judged profile artefacts, written under gpurun_out/r05/ (copied into profiles/ afterwards)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VDN_LIB_FLAVOUR=testing      # (round 6: the VDN_* switches below exist in the testing build of the library only)
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
# 1. kernel trace + stats of the default bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra --no-pmc > $O/bench.log 2>&1
# 2. the same command, sqlite, for the gap analysis
rocprofv3 --kernel-trace -d $O/bench_db -o t -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra --no-pmc > $O/bench_db.log 2>&1
python tools/trace_gaps.py $O/bench_db/t_results.db 5 grid > $O/trace_gaps.txt 2>&1
# 3. HBM counters of the whole bench (separate passes)
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr TA_BUSY_max" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/p$i -o p -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --no-extra --no-pmc > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
# 4. the smoother probe (the roofline kernel alone): counters per launch + copy calibration
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr TA_BUSY_max" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/smo/p$i -o p -- python3 tools/smoother_probe.py 256 20 > $O/smo$i.log 2>&1
done
python tools/pmc_summary.py $O/smo > $O/smoother_pmc_summary.txt 2>&1
python tools/make_pmc_json.py $O/smoother_pmc_summary.txt $O/smoother_split_pmc.json > /dev/null 2>&1
# 4b. the interleaved pass (VDN_MAC_SPLIT=0; what every multi-box / multi-rank level runs): fetch and write sizes
export VDN_MAC_SPLIT=0
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr TA_BUSY_max" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/smo_pair/p$i -o p -- python3 tools/smoother_probe.py 256 20 > $O/smo_pair$i.log 2>&1
done
unset VDN_MAC_SPLIT
python tools/pmc_summary.py $O/smo_pair > $O/smoother_pair_pmc_summary.txt 2>&1
python tools/make_pmc_json.py $O/smoother_pair_pmc_summary.txt $O/smoother_rho_pmc.json pair > /dev/null 2>&1
ls $O; head -n 12 $O/smoother_pmc_summary.txt; tail -n 1 $O/bench.log | cut -c1-300
