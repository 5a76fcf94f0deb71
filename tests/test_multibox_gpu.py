"""Domain decomposition on ONE GPU: the same domain cut into 2x2x2 boxes (all owned by rank 0, so every
box-to-box path is exercised except the RCCL wire) must reproduce the single-box results: ghost exchange
bit-exactly, the multigrid solves to 1e-11 (they run the same global hierarchy: per-box levels + an
agglomerated tail), the Godunov kernels up to the per-box dead-band eps (reference quirk, velpred.f90:1965-1980:
eps is relative to the box's own max velocity), i.e. to round-off."""
import ctypes as C

import numpy as np
import pytest

from tests.util import BC_SETS, Case, assert_bits

pytestmark = pytest.mark.gpu


class Split:
    """a Case's domain cut into nb[0] x nb[1] x nb[2] equal boxes"""

    def __init__(self, case, nb=(2, 2, 2)):
        from varden_amd import boxlib as bl
        self.case, self.nb = case, nb
        n = case.n
        self.bs = tuple(n[d] // nb[d] for d in range(3))
        self.boxes = []
        for kz in range(nb[2]):
            for ky in range(nb[1]):
                for kx in range(nb[0]):
                    lo = (kx * self.bs[0], ky * self.bs[1], kz * self.bs[2])
                    self.boxes.append((lo, tuple(lo[d] + self.bs[d] - 1 for d in range(3))))
        self.mla = bl.MLLayout([(case.lo, case.hi)], [self.boxes], pmask=case.pmask)
        self.bct = bl.BCTower(self.mla, case.phys)
        self.mfs = []

    def scatter(self, ofab, poison_ghosts=False):
        """multifab on the split layout holding the global fab's data (valid + whatever ghosts the global fab has)"""
        from varden_amd import boxlib as bl
        mf = bl.MultiFab(self.mla, 0, ofab.nc, ofab.ng, ofab.nodal)
        ng = ofab.ng
        for i, (lo, hi) in enumerate(self.boxes):
            sl = tuple(slice(lo[d], hi[d] + 1 + ofab.nodal[d] + 2 * ng) for d in range(3))
            a = np.array(ofab.a[sl], order="F")
            if poison_ghosts and ng:
                b = np.full_like(a, np.nan)
                b[ng:-ng, ng:-ng, ng:-ng] = a[ng:-ng, ng:-ng, ng:-ng]
                a = b
            mf.from_numpy(a, i)
        self.mfs.append(mf)
        return mf

    def gather(self, mf, ofab_like):
        """global array assembled from the boxes' VALID points (nodal duplicates: last writer wins)"""
        out = np.full(ofab_like.a.shape, np.nan, order="F")
        ng = mf.ng
        for i, (lo, hi) in enumerate(self.boxes):
            a = mf.to_numpy(i)
            v = a[ng:a.shape[0] - ng, ng:a.shape[1] - ng, ng:a.shape[2] - ng] if ng else a
            sl = tuple(slice(lo[d] + ng, lo[d] + ng + v.shape[d]) for d in range(3))
            out[sl] = v
        return out

    def close(self):
        for m in self.mfs:
            m.destroy()
        self.bct.destroy()
        self.mla.destroy()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "mixed"])
def test_fill_boundary_multibox(gpu, oracle, bcname):
    case = Case((16, 16, 16), BC_SETS[bcname], seed=21)
    sp = Split(case)
    for nodal, ng, nc in (((0, 0, 0), 3, 2), ((1, 0, 0), 1, 1), ((0, 0, 1), 1, 1), ((1, 1, 1), 1, 1)):
        g = case.ofab(ng, nc, nodal)
        g.a[...] = case.rng.standard_normal(g.a.shape)
        for d in range(3):
            if nodal[d] and case.pmask[d]:
                hi_sl, lo_sl = [slice(None)] * 4, [slice(None)] * 4
                hi_sl[d], lo_sl[d] = -(ng + 1), ng
                g.a[tuple(hi_sl)] = g.a[tuple(lo_sl)]
        oracle.lib().vo_fill_boundary(g.ref, case.opm)
        mf = sp.scatter(g, poison_ghosts=True)
        mf.fill_boundary()
        for i, (lo, hi) in enumerate(sp.boxes):
            a = mf.to_numpy(i)
            sl = tuple(slice(lo[d], hi[d] + 1 + nodal[d] + 2 * ng) for d in range(3))
            want = g.a[sl]
            # ghost points outside a non-periodic domain are not touched by fill_boundary: ignore them
            inside = np.ones(a.shape[:3], dtype=bool)
            for d in range(3):
                if not case.pmask[d]:
                    idx = np.arange(lo[d] - ng, hi[d] + 1 + nodal[d] + ng)
                    ok = (idx >= 0) & (idx <= case.hi[d] + nodal[d])
                    shp = [1, 1, 1]; shp[d] = -1
                    inside &= ok.reshape(shp)
            assert np.array_equal(a[inside], want[inside]), "fill_boundary nodal=%r box %d (%s)" % (nodal, i, bcname)
    sp.close(); case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout"])
def test_cc_solve_multibox_equals_single_box(gpu, oracle, bcname):
    from varden_amd import advance as adv
    case = Case((32, 32, 32), BC_SETS[bcname], seed=22, iso=True)
    _, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    beta = [case.ofab(0, 1, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
    oracle.lib().vo_mk_mac_coeffs(s.ref, oracle.fab_ptr_array(beta))
    rh = case.ofab(0, 1)
    rh.a[...] = case.rng.standard_normal(rh.a.shape)
    ell = oracle.ellbc_of(case.obc)
    if all(ell[d][sd] != 1 for d in range(3) for sd in range(2)):
        rh.a[...] -= rh.a.mean()
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    # single box
    gphi1 = case.gmf(case.ofab(1, 1))
    c1 = adv.cc_solve(case.gmf(rh), gphi1, [case.gmf(b) for b in beta], case.dx, bc, 1e-10)
    # 2x2x2 boxes
    sp = Split(case)
    gphi8 = sp.scatter(case.ofab(1, 1))
    c8 = adv.cc_solve(sp.scatter(rh), gphi8, [sp.scatter(b) for b in beta], case.dx, bc, 1e-10)
    assert c8[0] == c1[0] and c8[1] == c1[1], (c1, c8)
    one = gphi1.to_numpy()
    eight = sp.gather(gphi8, case.ofab(1, 1))
    a, b = eight[1:-1, 1:-1, 1:-1, 0], one[1:-1, 1:-1, 1:-1, 0]
    scale = np.abs(b - b.mean()).max()
    assert np.abs(a - b).max() <= 1e-11 * scale, (np.abs(a - b).max(), scale)
    sp.close(); case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic"])
def test_godunov_multibox(gpu, oracle, bcname):
    from varden_amd import advance as adv
    case = Case((16, 16, 16), BC_SETS[bcname], seed=23)
    u, s = case.random_state()
    force = case.ofab(1, 3)
    force.a[...] = case.rng.standard_normal(force.a.shape)
    oracle.lib().vo_fill_boundary(force.ref, case.opm)
    dt = 0.4 * min(case.dx)
    faces = lambda ng, nc, val=0.0: [case.ofab(ng, nc, tuple(1 if t == d else 0 for t in range(3)), val) for d in range(3)]  # noqa: E731
    # single box
    um1 = [case.gmf(f) for f in faces(1, 1, 1e20)]
    adv.velpred(case.gmf(u), um1, case.gmf(force), case.dx, dt, case.bct)
    sp = Split(case)
    um8 = [sp.scatter(f) for f in faces(1, 1, 1e20)]
    adv.velpred(sp.scatter(u), um8, sp.scatter(force), case.dx, dt, sp.bct)
    for d in range(3):
        a = sp.gather(um8[d], faces(1, 1)[d])[1:-1, 1:-1, 1:-1]
        b = um1[d].to_numpy()[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max(), "umac[%d] multibox" % d
    sp.close(); case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout"])
def test_nd_solve_multibox_equals_single_box(gpu, oracle, bcname):
    from varden_amd import advance as adv
    case = Case((16, 16, 16), BC_SETS[bcname], seed=24, iso=True)
    L = oracle.lib()
    u, s = case.random_state()
    rhohalf = case.ofab(1, 1)
    rhohalf.a[...] = np.abs(s.a[2:-2, 2:-2, 2:-2, :1]) + 0.5
    gpz = case.ofab(1, 3)
    L.vo_create_uvec(u.ref, u.ref, rhohalf.ref, gpz.ref, C.c_double(1.0), C.byref(case.obc), 1)
    L.vo_fill_boundary(u.ref, case.opm)
    coeffs = case.ofab(1, 1)
    coeffs.a[1:-1, 1:-1, 1:-1, 0] = 1.0 / rhohalf.a[1:-1, 1:-1, 1:-1, 0]
    L.vo_fill_boundary(coeffs.ref, case.opm)
    ell = oracle.ellbc_of(case.obc)
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    nodal = (1, 1, 1)
    p1 = case.gmf(case.ofab(1, 1, nodal))
    c1 = adv.nd_solve(case.gmf(case.ofab(1, 1, nodal)), p1, case.gmf(coeffs), case.gmf(u), case.dx, bc, 1e-11)
    sp = Split(case)
    p8 = sp.scatter(case.ofab(1, 1, nodal))
    c8 = adv.nd_solve(sp.scatter(case.ofab(1, 1, nodal)), p8, sp.scatter(coeffs), sp.scatter(u), case.dx, bc, 1e-11)
    assert c8[0] == c1[0] and c8[1] == c1[1], (c1, c8)
    a = sp.gather(p8, case.ofab(1, 1, nodal))[1:-1, 1:-1, 1:-1, 0]
    b = p1.to_numpy()[1:-1, 1:-1, 1:-1, 0]
    scale = np.abs(b - b.mean()).max()
    assert np.abs(a - b).max() <= 1e-11 * scale, (float(np.abs(a - b).max()), float(scale))
    sp.close(); case.close()


@pytest.mark.parametrize("name,phys,prob", [("bubble-walls", BC_SETS["walls"], 1), ("bubble-periodic", BC_SETS["periodic"], 1),
                                            ("blob-inout", BC_SETS["inout"], 2)])
def test_advance_timestep_multibox_equals_single_box(gpu, name, phys, prob):
    """3 steps of the whole path on 2x2x2 boxes vs one box (tolerance 1e-9: the per-box eps of the Godunov
    kernels is decomposition dependent by the reference's own definition)"""
    from tests.util import params_for
    from varden_amd import driver
    n = 32
    res = []
    for decomp in ((1, 1, 1), (2, 2, 2)):
        G = driver.Varden(n, phys, params_for(phys, cflfac=0.9), prob_type=prob, init_shrink=0.1, init_iter=1, decomp=decomp)
        for _ in range(3):
            G.step()
        res.append((G.gather_valid(G.unew[0]), G.gather_valid(G.snew[0]), G.dt))
        G.close()
    (u1, s1, dt1), (u8, s8, dt8) = res
    assert abs(dt1 - dt8) <= 1e-12 * dt1
    assert np.abs(u1 - u8).max() <= 1e-9 * max(np.abs(u1).max(), 1e-300), float(np.abs(u1 - u8).max())
    assert np.abs(s1 - s8).max() <= 1e-9 * np.abs(s1).max()


def test_packed_exchange_path(gpu, oracle, monkeypatch):
    """VDN_FORCE_PACKED=1 routes the rank's own box-to-box copies through the pack -> buffer -> unpack kernels
    that cross-rank copies use (only the ncclSend/ncclRecv hand-over is replaced by a device memcpy)"""
    monkeypatch.setenv("VDN_FORCE_PACKED", "1")
    test_fill_boundary_multibox(gpu, oracle, "periodic")
    test_fill_boundary_multibox(gpu, oracle, "mixed")
    test_cc_solve_multibox_equals_single_box(gpu, oracle, "periodic")
    test_nd_solve_multibox_equals_single_box(gpu, oracle, "walls")
    test_advance_timestep_multibox_equals_single_box(gpu, "bubble-periodic", BC_SETS["periodic"], 1)


def test_rccl_self_exchange_path(gpu, oracle, monkeypatch):
    """VDN_FORCE_PACKED=2 on ONE GPU: a 1-rank RCCL communicator is built through the same dlopen'ed entry points the
    multi-rank job uses, the rank's own packed halo buffers travel through ncclSend/ncclRecv (inside one group) and
    the norms through ncclAllReduce / ncclAllGather -- the hardware check of the RCCL call sequence that a one-GPU
    box allows (two ranks on one device are refused by RCCL)"""
    monkeypatch.setenv("VDN_FORCE_PACKED", "2")
    gpu.comm_init(gpu.comm_get_unique_id())
    try:
        test_fill_boundary_multibox(gpu, oracle, "periodic")
        test_fill_boundary_multibox(gpu, oracle, "mixed")
        test_cc_solve_multibox_equals_single_box(gpu, oracle, "periodic")
        test_nd_solve_multibox_equals_single_box(gpu, oracle, "walls")
        test_advance_timestep_multibox_equals_single_box(gpu, "bubble-periodic", BC_SETS["periodic"], 1)
    finally:
        gpu.comm_finalize()
