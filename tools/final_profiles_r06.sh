#!/bin/bash
# the round's judged profile artefacts, written under gpurun_out/r06/ (copied into profiles/ afterwards).  Part 1 (this call): the default bench line with its
# in-run counters, kernel stats + trace of the same command, the 512^3-in-eight-boxes kernel stats.  Part 2 (arg "pmc"): the counter passes.
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
if [ "$1" != "pmc" ]; then
  # 0. the default line, exactly as the driver runs it
  python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default_line.json; tail -c 300 $O/bench_default.log; echo
  # 1. kernel trace + stats of the headline workload
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra --no-pmc > $O/bench.log 2>&1
  cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
  # 2. the same command, sqlite, for the gap analysis
  rocprofv3 --kernel-trace -d $O/bench_db -o t -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --no-extra --no-pmc > $O/bench_db.log 2>&1
  python tools/trace_gaps.py $(find $O/bench_db -name "*results.db" | head -1) 5 grid > $O/r06_bench_trace_gaps.txt 2>&1
  # 3. configs[2] on one GPU: eight boxes (no one-box companion run inside the trace)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/b512 -o b -- python3 bench.py --config 512 --steps 3 --warmup 1 --skip-cpu --no-pmc --no-extra > $O/b512.log 2>&1
  cp $(find $O/b512 -name "*kernel_stats.csv" | head -1) $O/r06_bench512_kernel_stats.csv
  head -5 $O/r06_bench_trace_gaps.txt | cut -c1-200
else
  # 4. HBM counters of the whole bench (separate passes)
  i=0
  for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr TA_BUSY_max" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
    i=$((i+1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/p$i -o p -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --no-extra --no-pmc > $O/pmc$i.log 2>&1
  done
  python tools/pmc_summary.py $O/pmc > $O/r06_bench_pmc_summary.txt 2>&1
  # 5. the smoother probe (the roofline kernel alone): counters per launch + copy calibration
  i=0
  for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr TA_BUSY_max" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitBusy"; do
    i=$((i+1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/smo/p$i -o p -- python3 tools/smoother_probe.py 256 20 > $O/smo$i.log 2>&1
  done
  python tools/pmc_summary.py $O/smo > $O/r06_smoother_pmc_summary.txt 2>&1
  python tools/make_pmc_json.py $O/r06_smoother_pmc_summary.txt $O/r06_smoother_split_pmc.json > /dev/null 2>&1
  head -n 8 $O/r06_smoother_pmc_summary.txt | cut -c1-220
fi
