"""the tagged box lists of the bubble (tag_boxes.f90:65-94 + make_new_grids) as JSON -- input of oracle-side experiments on the CPU
usage: dump_tagged_grids.py <base> <max_levs> <max_grid_size> <out.json>"""
import json
import sys
sys.path.insert(0, ".")
from varden_amd import driver
from varden_amd.capi import default_params
nc, ml, mgs, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
levels = driver.VardenAMR.tagged_grids(nc, [[15, 15]] * 3, default_params(cflfac=0.9), max_levs=ml, max_grid_size=mgs)
json.dump([[[list(b[0]), list(b[1])] for b in lb] for lb in levels], open(out, "w"))
print(nc, [len(lb) for lb in levels])
