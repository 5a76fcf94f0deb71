#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests/test_multibox_gpu.py tests/test_multirank_gpu.py tests/test_projection_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/r06/agglom_tests.log 2>&1; echo "rc $?" >> gpurun_out/r06/agglom_tests.log; tail -15 gpurun_out/r06/agglom_tests.log
python -m pytest tests/test_fullsize_gpu.py -x -q -k "512" >> gpurun_out/r06/agglom_tests.log 2>&1; echo "rc $?" >> gpurun_out/r06/agglom_tests.log; tail -5 gpurun_out/r06/agglom_tests.log
python bench.py --config 512 --steps 5 --warmup 2 --skip-cpu --no-pmc --no-extra > gpurun_out/r06/bench512_agglom.log 2>&1; grep '^{' gpurun_out/r06/bench512_agglom.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['phase_ms_per_step'], d['config']['vcycles_per_step'])"
