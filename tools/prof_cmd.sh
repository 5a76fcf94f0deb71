#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: prof_cmd.sh tag script args...   -> gpurun_out/<tag>.txt (per-kernel totals)
tag=$1; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o t -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python - "$tag" <<'PY' > gpurun_out/$tag.txt
import csv, sys, glob, collections
tag = sys.argv[1]
f = glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %5d %10.1f us total %9.1f us avg" % (k, v[0], v[1], v[1] / v[0]))
PY
head -n 30 gpurun_out/$tag.txt
