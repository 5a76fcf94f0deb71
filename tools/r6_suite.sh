#!/bin/bash
# the whole GPU suite, log under gpurun_out/r06 (a progress line per test file keeps the call alive)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q --deselect "tests/test_fullsize_gpu.py::test_tagged_hierarchy_step_at_256" "$@" > gpurun_out/r06/suite.log 2>&1
rc=$?
tail -15 gpurun_out/r06/suite.log
exit $rc
