#!/bin/bash
# the default bench line's extra workloads (256^3, then 512^3 in eight boxes, then ONE 512^3 box, ...) with pooled-chunk fields and with hipMalloc fields: the one-box case after the pool has been churned
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/alloc_ab
run() { tag=$1; shift; env VDN_LIB_FLAVOUR=testing "$@" python bench.py --skip-cpu --no-pmc > gpurun_out/alloc_ab/$tag.json 2> gpurun_out/alloc_ab/$tag.err || return 1
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/alloc_ab/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-24s 256^3 %.2f ms | " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%.1f" % e["ms_per_step"] for e in d["extra_workloads"]), flush=True)
PY
}
run line_default && run line_default_again
