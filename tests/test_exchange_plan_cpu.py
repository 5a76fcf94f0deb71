"""Host logic of the N > 1 path, on CPU with 2 gloo ranks: each rank asks the library (vdn_plan_describe, pure
host code) which remote ghost copies it must send / receive for a 2x2x2 decomposition, executes that plan with
numpy pack/unpack + torch.distributed(gloo) send/recv, fills the local copies itself, and the result must equal
the ghost cells cut out of the globally filled array.  This pins the canonical (dst, src, shift) enumeration
and the buffer offsets that both sides of an RCCL send/recv pair rely on (exchange.hip)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def global_fill(a, ng, nodal, pmask, n):
    """reference result: ghost points of the whole-domain array filled by periodic wrap (None where not periodic)"""
    out = a.copy()
    idx = []
    for d in range(3):
        i = np.arange(-ng, n[d] + nodal[d] + ng)
        if pmask[d]:
            src = np.where(i < 0, i + n[d], np.where(i > n[d] - 1 + nodal[d], i - n[d], i))
        else:
            src = np.clip(i, 0, n[d] - 1 + nodal[d])
        idx.append(src + ng)
    return out[np.ix_(idx[0], idx[1], idx[2])]


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from varden_amd import capi
    from varden_amd.capi import Box
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = capi.load()
    try:
        for pmask, nodal, ng, nc in (((1, 1, 1), (0, 0, 0), 3, 2), ((0, 1, 0), (1, 0, 0), 1, 1), ((1, 0, 1), (1, 1, 1), 1, 1)):
            n, bs = (8, 8, 8), (4, 4, 4)
            boxes = [((kx * 4, ky * 4, kz * 4), (kx * 4 + 3, ky * 4 + 3, kz * 4 + 3)) for kz in range(2) for ky in range(2) for kx in range(2)]
            owner = [i % world for i in range(8)]
            rng = np.random.default_rng(7)
            gshape = tuple(n[d] + nodal[d] + 2 * ng for d in range(3)) + (nc,)
            glob = rng.standard_normal(gshape)
            for d in range(3):      # nodal duplicates on periodic faces must agree
                if nodal[d] and pmask[d]:
                    hi_sl, lo_sl = [slice(None)] * 4, [slice(None)] * 4
                    hi_sl[d], lo_sl[d] = -(ng + 1), ng
                    glob[tuple(hi_sl)] = glob[tuple(lo_sl)]
            want = np.stack([global_fill(glob[..., c], ng, nodal, pmask, n) for c in range(nc)], axis=-1)
            # local boxes: valid data from the global array, ghosts poisoned
            loc = {}
            for g, (lo, hi) in enumerate(boxes):
                if owner[g] != rank:
                    continue
                sl = tuple(slice(lo[d], hi[d] + 1 + nodal[d] + 2 * ng) for d in range(3))
                a = np.full_like(glob[sl], np.nan)
                a[ng:-ng, ng:-ng, ng:-ng] = glob[sl][ng:-ng, ng:-ng, ng:-ng]
                loc[g] = a
            pd = Box(); pd.lo[:] = (0, 0, 0); pd.hi[:] = (7, 7, 7)
            barr = (Box * 8)()
            for g, (lo, hi) in enumerate(boxes):
                barr[g].lo[:] = lo; barr[g].hi[:] = hi
            rows = (C.c_long * (14 * 4096))()
            nrows, nloc = C.c_int(), C.c_int()
            rc = lib.vdn_plan_describe(C.byref(pd), (C.c_int * 3)(*pmask), 8, barr, (C.c_int * 8)(*owner), nc, ng, (C.c_int * 3)(*nodal), rank,
                                       rows, 4096, C.byref(nrows), C.byref(nloc))
            assert rc == 0 and nrows.value <= 4096
            R = np.array(rows[:14 * nrows.value]).reshape(-1, 14)

            def view(g, lo, hi, sh=(0, 0, 0)):      # slice of box g's array for the index box [lo-sh, hi-sh]
                blo = boxes[g][0]
                return tuple(slice(lo[d] - sh[d] - blo[d] + ng, hi[d] - sh[d] - blo[d] + ng + 1) for d in range(3))

            def ghost_mask(g, lo, hi):              # True where the point is NOT a valid point of box g
                blo, bhi = boxes[g]
                m = np.zeros(tuple(hi[d] - lo[d] + 1 for d in range(3)), dtype=bool)
                for d in range(3):
                    i = np.arange(lo[d], hi[d] + 1)
                    o = (i < blo[d]) | (i > bhi[d] + nodal[d])
                    shp = [1, 1, 1]; shp[d] = -1
                    m |= o.reshape(shp)
                return m

            # pack + exchange (buffer layout of k_xpack: per descriptor, component-major, x fastest)
            sendbuf, recvn = {}, {}
            for r in R:
                kind, peer, lo, hi, sh, off, gd, gs = r[0], int(r[1]), r[2:5], r[5:8], r[8:11], r[11], int(r[12]), int(r[13])
                cnt = int(np.prod(hi - lo + 1)) * nc
                if kind == 0:
                    blk = loc[gs][view(gs, lo, hi, sh)]
                    buf = sendbuf.setdefault(peer, {})
                    buf[int(off)] = np.concatenate([blk[..., c].ravel(order="F") for c in range(nc)])
                else:
                    recvn[peer] = max(recvn.get(peer, 0), int(off) + cnt)
            reqs, rbuf = [], {}
            for peer in sorted(set(list(sendbuf) + list(recvn))):
                if peer in sendbuf:
                    flat = np.concatenate([sendbuf[peer][o] for o in sorted(sendbuf[peer])])
                    reqs.append(dist.isend(torch.from_numpy(flat), peer))
                if peer in recvn:
                    rbuf[peer] = torch.empty(recvn[peer], dtype=torch.float64)
                    reqs.append(dist.irecv(rbuf[peer], peer))
            for q_ in reqs:
                q_.wait()
            for r in R:
                kind, peer, lo, hi, sh, off, gd, gs = r[0], int(r[1]), r[2:5], r[5:8], r[8:11], int(r[11]), int(r[12]), int(r[13])
                if kind != 1:
                    continue
                shp = tuple(hi - lo + 1)
                tot = int(np.prod(shp))
                m = ghost_mask(gd, lo, hi)
                for c in range(nc):
                    blk = rbuf[peer][off + c * tot: off + (c + 1) * tot].numpy().reshape(shp, order="F")
                    tgt = loc[gd][view(gd, lo, hi)][..., c]
                    tgt[m] = blk[m]
            # local copies: the same enumeration restricted to boxes I own
            per = n
            for gd in loc:
                for gs in loc:
                    for sz in ([-1, 0, 1] if pmask[2] else [0]):
                        for sy in ([-1, 0, 1] if pmask[1] else [0]):
                            for sx in ([-1, 0, 1] if pmask[0] else [0]):
                                if gd == gs and (sx, sy, sz) == (0, 0, 0):
                                    continue
                                sh = (sx * per[0], sy * per[1], sz * per[2])
                                lo = [max(boxes[gd][0][d] - ng, boxes[gs][0][d] + sh[d]) for d in range(3)]
                                hi = [min(boxes[gd][1][d] + nodal[d] + ng, boxes[gs][1][d] + nodal[d] + sh[d]) for d in range(3)]
                                if any(lo[d] > hi[d] for d in range(3)):
                                    continue
                                m = ghost_mask(gd, lo, hi)
                                src = loc[gs][view(gs, lo, hi, sh)]
                                tgt = loc[gd][view(gd, lo, hi)]
                                tgt[m] = src[m]
            # check: every ghost point that has a source (inside the domain or periodic) equals the global fill
            for g, a in loc.items():
                lo, hi = boxes[g]
                sl = tuple(slice(lo[d], hi[d] + 1 + nodal[d] + 2 * ng) for d in range(3))
                w = want[sl]
                ok = np.ones(a.shape[:3], dtype=bool)
                for d in range(3):
                    if not pmask[d]:
                        i = np.arange(lo[d] - ng, hi[d] + 1 + nodal[d] + ng)
                        o = (i >= 0) & (i <= n[d] - 1 + nodal[d])
                        shp = [1, 1, 1]; shp[d] = -1
                        ok &= o.reshape(shp)
                assert np.array_equal(a[ok], w[ok]), "rank %d box %d pmask %r nodal %r" % (rank, g, pmask, nodal)
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_exchange_plan_two_gloo_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for r, msg in res:
        assert msg == "ok", "rank %d: %s" % (r, msg)


def test_box_index_names_every_touching_box():
    """the bin index behind the box-pair loops of the inter-level operators (vdn_internal.h BoxBins, host code): for random box lists -- sizes 1 to 40, some
    far outside the cloud, negative indices -- and random query regions its candidates are ascending, unique, and contain every box that touches the grown query"""
    from varden_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(5)
    for trial in range(40):
        nb = int(rng.integers(1, 400))
        los = rng.integers(-64, 512, size=(nb, 3)); ext = rng.integers(0, 40, size=(nb, 3))
        if trial % 5 == 0:
            los[0] = (-5000, 3000, 7); ext[0] = (0, 0, 0)
        boxes = (capi.Box * nb)()
        for i in range(nb):
            for d in range(3):
                boxes[i].lo[d] = int(los[i, d]); boxes[i].hi[d] = int(los[i, d] + ext[i, d])
        for q in range(30):
            qlo = rng.integers(-100, 560, size=3); qhi = qlo + rng.integers(0, 70, size=3)
            margin = int(rng.integers(0, 4))
            out = (C.c_int * nb)(); nc = C.c_int(0)
            rc = lib.vdn_box_candidates(nb, boxes, (C.c_int * 3)(*[int(x) for x in qlo]), (C.c_int * 3)(*[int(x) for x in qhi]), margin, out, nb, C.byref(nc))
            assert rc == 0
            cand = list(out[:nc.value])
            assert cand == sorted(set(cand)) and all(0 <= c < nb for c in cand)
            touch = np.all((los + ext >= qlo - margin) & (los <= qhi + margin), axis=1)
            assert set(np.nonzero(touch)[0]) <= set(cand), (trial, q)
