set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_projection_gpu.py -x -q -k "split_colour" > gpurun_out/r6a_pytest_split.log 2>&1
echo "rc split $?" >> gpurun_out/r6a_pytest_split.log
tail -5 gpurun_out/r6a_pytest_split.log
python -m pytest tests/test_multirank_gpu.py -x -q -k "by_colour or default_overlap" > gpurun_out/r6a_pytest_mr.log 2>&1
echo "rc mr $?" >> gpurun_out/r6a_pytest_mr.log
tail -5 gpurun_out/r6a_pytest_mr.log
python bench.py --config 512 --steps 5 --warmup 2 --skip-cpu --no-pmc > gpurun_out/r6a_bench512.log 2>&1
tail -c 1500 gpurun_out/r6a_bench512.log
VDN_MAC_SPLIT_HALO=0 python bench.py --config 512 --steps 5 --warmup 2 --skip-cpu --no-pmc > gpurun_out/r6a_bench512_r5form.log 2>&1
tail -c 600 gpurun_out/r6a_bench512_r5form.log
