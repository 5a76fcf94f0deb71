export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/packed; mkdir -p $O
export VDN_FORCE_PACKED=2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov1 -o p -- python3 bench.py --config 512 --steps 2 --warmup 1 --skip-cpu > $O/prof_ov1.log 2>&1
export VDN_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov0 -o p -- python3 bench.py --config 512 --steps 2 --warmup 1 --skip-cpu > $O/prof_ov0.log 2>&1
find $O -name "*kernel_stats.csv" | head
