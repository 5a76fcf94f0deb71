#!/bin/bash
# rocprofv3 kernel statistics of one bench configuration: prof_cfg.sh <tag> <bench args...> -> gpurun_out/<tag>/
tag=$1; shift; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o a -- python3 bench.py "$@" --skip-cpu > gpurun_out/$tag.log 2>&1
tail -n 1 gpurun_out/$tag.log | cut -c1-200
