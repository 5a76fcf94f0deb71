export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/packed; mkdir -p $O
python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu > $O/local.log 2>&1; tail -n 1 $O/local.log | cut -c1-400
VDN_FORCE_PACKED=2 timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu > $O/packed2.log 2>&1; tail -n 1 $O/packed2.log | cut -c1-400
VDN_FORCE_PACKED=2 VDN_OVERLAP=0 timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu > $O/packed2_noov.log 2>&1; tail -n 1 $O/packed2_noov.log | cut -c1-400
VDN_FORCE_PACKED=1 timeout -k 10 300 python bench.py --config 512 --steps 3 --warmup 1 --skip-cpu > $O/packed1.log 2>&1; tail -n 1 $O/packed1.log | cut -c1-400
