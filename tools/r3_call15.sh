cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3c15; mkdir -p $O
for m in 0 1; do
VDN_MAC_FAST=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$m -o b -- python3 bench.py --config 512 --steps 2 --warmup 1 --skip-cpu > $O/p$m.log 2>&1
f=$(find $O/p$m -name "*kernel_stats.csv" | head -n 1); cp "$f" $O/stats_macfast$m.csv
done
echo done
