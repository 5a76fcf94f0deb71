#!/bin/bash
# the GPU suite with every byte handed back to the arena overwritten with NaNs (VDN_ARENA_POISON=1, testing build -- tests/conftest.py selects it): a kernel that reads an arena entry
# nobody wrote meets a NaN and fails the next solve.  Log under gpurun_out/r06.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
VDN_ARENA_POISON=1 python -m pytest tests -m gpu -x -q --durations=10 "$@" > gpurun_out/r06/suite_poison.log 2>&1
rc=$?
tail -15 gpurun_out/r06/suite_poison.log
exit $rc
