#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests/test_fullsize_gpu.py -x -q -k "tagged_hierarchy_step_at_256" > gpurun_out/r06/fixture_test.log 2>&1; echo "rc $?" >> gpurun_out/r06/fixture_test.log; tail -15 gpurun_out/r06/fixture_test.log
python -m pytest tests/test_amr_gpu.py -x -q -k "advect-periodic-x" > gpurun_out/r06/periodic_test.log 2>&1; echo "rc $?" >> gpurun_out/r06/periodic_test.log; tail -25 gpurun_out/r06/periodic_test.log
bash tools/r6_gpu_batch.sh
