#!/bin/bash
# bench.py --config 512 (eight 256^3 boxes, then one 512^3 box) under the allocator variants of runtime.hip (testing build): pooled 64 MB / 1 GB chunks behind the state fields, plain hipMalloc fields,
# 256 MB arena chunks -- does the mapping granularity cost bandwidth (TLB reach)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/alloc_ab
run() { tag=$1; shift; env VDN_LIB_FLAVOUR=testing "$@" python bench.py --config 512 --skip-cpu --no-pmc --steps 10 --warmup 2 > gpurun_out/alloc_ab/$tag.json 2> gpurun_out/alloc_ab/$tag.err || return 1
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/alloc_ab/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s eight boxes %.2f ms   one 512^3 box %.2f ms" % (sys.argv[1], d["ms_per_step"], d["config"]["one_box_512"]["ms_per_step"]), flush=True)
PY
}
run default_64MB_fields && run fields_hipMalloc VDN_FIELD_VMM=0 && run fields_1GB_chunks VDN_FIELD_CHUNK_MB=1024 && run arena_256MB_chunks VDN_ARENA_CHUNK_MB=256 && run default_again
