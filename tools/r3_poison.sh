#!/bin/bash
# the GPU suite with every byte handed back to the arena overwritten by NaNs (VDN_ARENA_POISON=1, runtime.hip): a kernel that reads an entry
# nobody wrote -- and worked because the previous tenant of the address left zeros -- fails a solve instead of passing by luck
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
VDN_ARENA_POISON=1 timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/pytest_poison.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_poison.log
tail -n 8 gpurun_out/pytest_poison.log | cut -c1-200
