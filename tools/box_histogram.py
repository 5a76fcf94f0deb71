import sys, collections
sys.path.insert(0, ".")
from varden_amd import driver
from varden_amd.capi import default_params
walls = [[15, 15]] * 3
prm = default_params(cflfac=0.9)
levels = driver.VardenAMR.tagged_grids(256, walls, prm, max_levs=3, max_grid_size=256)
for n, lb in enumerate(levels):
    hx = collections.Counter(); hy = collections.Counter(); hz = collections.Counter(); cells = collections.Counter()
    for lo, hi in lb:
        w = [hi[d] - lo[d] + 1 for d in range(3)]
        hx[w[0]] += 1; hy[w[1]] += 1; hz[w[2]] += 1; cells[w[0]] += w[0] * w[1] * w[2]
    tot = sum(cells.values())
    print("level", n + 1, "boxes", len(lb), "cells", tot)
    print("  x widths (count, share of cells):", [(k, hx[k], round(cells[k] / tot, 3)) for k in sorted(hx)])
    print("  y widths:", sorted(hy.items()))
    print("  z widths:", sorted(hz.items()))
# cost model of the batched Godunov march (workgroup-planes: tiles in x and y times planes marched, a tile = 64 x 8 threads owning 62 x 6 cells), per class
# of x width, and what row segments of 16 / 32 lanes (own 14 / 30 cells, 32 / 16 rows per workgroup owning 30 / 14) would make of it
import math
for n, lb in enumerate(levels):
    cur = collections.Counter(); new = collections.Counter()
    for lo, hi in lb:
        w = [hi[d] - lo[d] + 1 for d in range(3)]
        cur[w[0]] += math.ceil(w[0] / 62) * math.ceil(w[1] / 6) * (w[2] + 4)
        best = None
        for W, rows in ((64, 8), (32, 16), (16, 32)):
            c = math.ceil(w[0] / (W - 2)) * math.ceil(w[1] / (rows - 2)) * (w[2] + 4)
            best = c if best is None else min(best, c)
        new[w[0]] += best
    tc, tn = sum(cur.values()), sum(new.values())
    ideal = sum((hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) for lo, hi in lb) / 512.0
    print("level", n + 1, "workgroup-planes now %d, with narrow segments %d (%.2f), cells/512 = %d" % (tc, tn, tn / tc, ideal))
    print("  share by x width now:", [(k, round(cur[k] / tc, 3)) for k in sorted(cur)])
    print("  share by x width then:", [(k, round(new[k] / tn, 3)) for k in sorted(new)])
# ... and with the number of waves per workgroup chosen per box as well (2, 4 or 8 waves; rows per workgroup = waves x rows per wave), in wave-planes
for n, lb in enumerate(levels):
    cur = 0; new = 0; new8 = 0
    for lo, hi in lb:
        w = [hi[d] - lo[d] + 1 for d in range(3)]
        cur += math.ceil(w[0] / 62) * math.ceil(w[1] / 6) * (w[2] + 4) * 8
        best = None; best8 = None
        for W in range(6, 65):
            rpw = 64 // W
            for nw in (2, 4, 8):
                if nw * rpw - 2 < 1: continue
                c = math.ceil(w[0] / (W - 2)) * math.ceil(w[1] / (nw * rpw - 2)) * (w[2] + 4) * nw
                best = c if best is None else min(best, c)
                if nw == 8: best8 = c if best8 is None else min(best8, c)
        new += best; new8 += best8
    ideal = sum((hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) for lo, hi in lb) / 64.0
    print("level", n + 1, "wave-planes now %d; any segment width, 8 waves: %d (%.2f); and 2 / 4 / 8 waves: %d (%.2f); cells/64 = %d" % (cur, new8, new8 / cur, new, new / cur, ideal))
# the box-batched cell kernels (vdn_dev.h batch_grid: 16 x 16 / 32 x 8 / 64 x 4 tiles by the width of the range, or the plane flattened over 256 threads when
# that takes fewer workgroups; ppw planes per workgroup).  GsrbB: the range is half a row wide (one colour), one plane per workgroup
def batch_wgs(nx, ny, nz, ppw=1):
    lw = 6 if nx > 32 else (5 if nx > 16 else 4); w = 1 << lw; h = 256 >> lw
    g = math.ceil(nx / w) * math.ceil(ny / h)
    g = min(g, math.ceil(nx * ny / 256))
    return g * math.ceil(nz / ppw)
for n, lb in enumerate(levels):
    gs = 0; rs = 0; ideal = 0; gs_by = collections.Counter()
    for lo, hi in lb:
        w = [hi[d] - lo[d] + 1 for d in range(3)]
        c = batch_wgs((w[0] + 1) // 2, w[1], w[2]); gs += c; gs_by[w[0]] += c
        rs += batch_wgs(w[0], w[1], w[2], 8) * 8
        ideal += w[0] * w[1] * w[2]
    print("level", n + 1, "colour pass: %d workgroups for %d needed (%.2f); residual: %d workgroup-planes for %d (%.2f)" % (gs, ideal / 512, gs / (ideal / 512), rs, ideal / 256, rs / (ideal / 256)))
    print("  colour pass workgroups by x width:", [(k, round(gs_by[k] / gs, 3)) for k in sorted(gs_by)])
# the nodal marches of the composite solve (mg_nd.hip ndf_build_march: a lane carries two nodes, row segments of sw lanes, 64 / sw rows per wave, four waves per
# workgroup, the whole box height per workgroup on levels of many boxes): thread-planes marched against node pairs
for n, lb in enumerate(levels):
    tp = 0; ideal = 0; by = collections.Counter()
    for lo, hi in lb:
        nn = [hi[d] - lo[d] + 2 for d in range(3)]
        best = None
        for sw in range(4, 65):
            cost = math.ceil(nn[0] / (2 * (sw - 2))) / (64 // sw)
            if best is None or cost < best[0] - 1e-12: best = (cost, sw)
        sw = best[1]; rows = 4 * (64 // sw)
        wgs = math.ceil(nn[0] / (2 * (sw - 2))) * math.ceil(nn[1] / rows)
        kchunk = nn[2]
        if not (len(lb) > 16 and nn[2] <= 64):
            while kchunk > 8 and wgs * math.ceil(nn[2] / kchunk) < 2048: kchunk = (kchunk + 1) // 2
        c = wgs * math.ceil(nn[2] / kchunk) * (kchunk + 2) * 256
        tp += c; by[hi[0] - lo[0] + 1] += c
        ideal += nn[0] * nn[1] * nn[2] / 2
    print("level", n + 1, "nodal march: %.2f thread-planes per node pair" % (tp / ideal), " share by x width:", [(k, round(by[k] / tp, 3)) for k in sorted(by)])
