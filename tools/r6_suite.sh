#!/bin/bash
# the whole GPU suite with the slowest tests listed, log under gpurun_out/r06
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q --durations=40 "$@" > gpurun_out/r06/suite.log 2>&1
rc=$?
tail -60 gpurun_out/r06/suite.log
exit $rc
