"""Golden vectors of BASELINE.json configs[3] / configs[4] AT FULL SIZE, written by the CPU oracle in the build container (no GPU needed):

    python tools/amr_fullsize_fixture.py 2      # -> tests/golden/amr2_fullsize_samples.npz   (256^3 base + 263 boxes: about 5 min on 8 cores)
    python tools/amr_fullsize_fixture.py 3 [out.npz]     # -> tests/golden/amr3_fullsize_samples.npz   (+ 997 boxes, 54.7 M cells: about an hour on 8 cores; the committed
                                                         #    file was written by this script on the 16 host cores of a GPU box, ten minutes -- it uses no GPU)

Input: the tagged box lists of the 256^3 bubble, tests/golden/amr_grids_256_l<max_levs>.json, dumped once on a GPU box by tools/dump_tagged_grids.py
(tag_boxes + make_new_grids run on the device; the GPU test asserts that the library still produces exactly these lists).  The oracle (oracle/voracle.py: SimML)
runs the start-up sequence and ONE step on them and the fixture keeps, per level: dt, the FAC iteration counts of both composite solves, per box the sums of
u, v, w, rho, tracer, gpx, gpy, gpz over its cells, and the values on every fourth cell of two planes through the level's bounding box.
tests/test_fullsize_gpu.py::test_tagged_hierarchy_fullsize_against_the_oracle_fixture steps the GPU on the same lists and compares at 1e-9 with equal counts --
the full-size parity of the hierarchies inside the driver's suite, no oracle run on the GPU box.  A few hundred KB per fixture."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ("u", "v", "w", "rho", "trac", "gpx", "gpy", "gpz")


def sample_planes(lo, hi, stride=4):
    """the cells (global indices of the level) of the fixture's two planes inside one box: z = mid-plane of `lo..hi` (the bounding box of the level) and x = its
    mid-plane, every `stride`-th cell of the in-plane directions (by GLOBAL index, so that neighbouring boxes continue the lattice)"""
    mid = [(lo[d] + hi[d]) // 2 for d in range(3)]
    return mid, stride


def box_samples(blo, bhi, mid, stride):
    out = []
    for nd in (2, 0):                                   # planes normal to z and to x
        if not (blo[nd] <= mid[nd] <= bhi[nd]):
            continue
        rng = [None, None, None]
        for d in range(3):
            if d == nd:
                rng[d] = np.array([mid[nd]])
            else:
                a = ((blo[d] + stride - 1) // stride) * stride
                rng[d] = np.arange(a, bhi[d] + 1, stride)
        I, J, K = np.meshgrid(*rng, indexing="ij")
        out.append(np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1))
    return np.concatenate(out, axis=0).astype(np.int32) if out else np.zeros((0, 3), np.int32)


def main():
    ml = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    from oracle import voracle as vo
    from varden_amd.capi import default_params
    vo.lib()
    nth = int(os.environ.get("OMP_NUM_THREADS", str(os.cpu_count() or 1)))
    vo.set_threads(nth)

    def beat():
        while True:
            time.sleep(60); print("  ... %s" % time.strftime("%H:%M:%S"), flush=True)
    threading.Thread(target=beat, daemon=True).start()
    nc, W = 256, [[15, 15]] * 3
    grids = json.load(open(os.path.join(ROOT, "tests", "golden", "amr_grids_256_l%d.json" % ml)))
    levels = [[(tuple(b[0]), tuple(b[1])) for b in lb] for lb in grids]
    t0 = time.time()
    O = vo.SimML(nc, levels, W, prm=default_params(cflfac=0.9), init_shrink=0.1, init_iter=1, do_initial_projection=1)
    print("oracle start-up %.1f s, dt %r, initial projection %r" % (time.time() - t0, O.dt, O.initial_projection_stat), flush=True)
    dt0 = O.dt
    t0 = time.time()
    O.step()
    print("oracle step %.1f s, FAC (MAC, HG) %r, dt %r" % (time.time() - t0, (O.mgstat[0].cycles, O.mgstat[1].cycles), O.dt), flush=True)
    out = {"nc": nc, "max_levs": ml, "dt_startup": dt0, "dt_step": O.dt, "fac_mac": int(O.mgstat[0].cycles), "fac_hg": int(O.mgstat[1].cycles),
           "fac_initial_projection": int(O.initial_projection_stat[0]), "names": np.array(NAMES)}
    for n in range(O.nlev):
        olo = O.levels[n].lo
        boxes = [((0, 0, 0), (nc - 1,) * 3)] if n == 0 else levels[n - 1]
        blo = [min(b[0][d] for b in boxes) for d in range(3)]
        bhi = [max(b[1][d] for b in boxes) for d in range(3)]
        mid, stride = sample_planes(blo, bhi)
        fields = [O.uold[n].valid()[..., c] for c in range(3)] + [O.sold[n].valid()[..., c] for c in range(2)] + [O.gp[n].valid()[..., c] for c in range(3)]
        sums = np.zeros((len(boxes), len(NAMES)))
        idx, val = [], []
        for bi, (lo, hi) in enumerate(boxes):
            sl = tuple(slice(lo[d] - olo[d], hi[d] - olo[d] + 1) for d in range(3))
            for c, f in enumerate(fields):
                sums[bi, c] = f[sl].sum()
            s = box_samples(lo, hi, mid, stride)
            if len(s):
                idx.append(s)
                val.append(np.stack([f[s[:, 0] - olo[0], s[:, 1] - olo[1], s[:, 2] - olo[2]] for f in fields], axis=1))
        out["boxsum_%d" % n] = sums
        out["idx_%d" % n] = np.concatenate(idx, axis=0)
        out["val_%d" % n] = np.concatenate(val, axis=0)
        out["absmax_%d" % n] = np.array([np.abs(f[O.levels[n].mask() != 0]).max() if hasattr(O.levels[n], "mask") else np.abs(f).max() for f in fields])
        print("level %d: %d boxes, %d sampled cells" % (n, len(boxes), len(out["idx_%d" % n])), flush=True)
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "amr%d_fullsize_samples.npz" % ml)      # (an explicit path: run elsewhere, copy back)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
