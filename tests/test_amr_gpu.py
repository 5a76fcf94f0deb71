"""Two-level AMR (BASELINE.json configs[3]): FBoxLib's multi-level operators as defined in oracle/vo_amr.c (the reference calls
them but does not contain them) and the multilevel MAC projection (src/macproject.f90:20-133), HIP vs oracle.
Transfer operators: bit-exact.  Composite solve: same FAC iteration count, MAC velocities to 1e-9, and the size-independent
property that the projected field is discretely divergence-free on the composite grid (fine cells, uncovered coarse cells,
and -- through ml_edge_restriction -- covered coarse cells)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import assert_bits

pytestmark = pytest.mark.gpu

WALLS = [[15, 15]] * 3


class Amr2:
    """coarse level: one box [0,nc)^3; fine level: the box flo..fhi (fine indices), optionally split in x for the GPU"""

    def __init__(self, nc, flo, fhi, phys=WALLS, split=1, seed=0, finer=(), fboxes=None, cboxes=None):
        """finer: (lo, hi) boxes of the levels 2.. (one box each, own index space); fboxes / cboxes: explicit GPU box lists of
        level 1 / level 0 (any partition of the oracle's one box)"""
        from oracle import voracle as vo
        from varden_amd import boxlib as bl
        from varden_amd.capi import default_params
        self.vo, self.bl = vo, bl
        self.nc, self.flo, self.fhi = nc, tuple(flo), tuple(fhi)
        self.prm = default_params()
        bl.initialize(self.prm, 0, 1, 0)
        self.rng = np.random.default_rng(seed)
        self.clo, self.chi = (0, 0, 0), (nc - 1,) * 3
        self.los = [self.clo, tuple(flo)] + [tuple(b[0]) for b in finer]
        self.his = [self.chi, tuple(fhi)] + [tuple(b[1]) for b in finer]
        self.nlev = NL = len(self.los)
        pds = [((0, 0, 0), ((nc << n) - 1,) * 3) for n in range(NL)]
        nx = (fhi[0] - flo[0] + 1) // split
        self.fboxes = [((flo[0] + s * nx, flo[1], flo[2]), (flo[0] + (s + 1) * nx - 1, fhi[1], fhi[2])) for s in range(split)]
        if fboxes is not None:
            self.fboxes = [(tuple(b[0]), tuple(b[1])) for b in fboxes]
        self.cboxes = [(self.clo, self.chi)] if cboxes is None else [(tuple(b[0]), tuple(b[1])) for b in cboxes]
        self.mla = bl.MLLayout(pds, [self.cboxes, self.fboxes] + [[(tuple(b[0]), tuple(b[1]))] for b in finer], rr=[(2, 2, 2)] * (NL - 1))
        self.bct = bl.BCTower(self.mla, phys)
        self.dx = [[1.0 / (nc << n)] * 3 for n in range(NL)]
        # oracle side: one box per level
        bcl, opd = [vo.make_bc(phys, 3, 2)], [0, 0, 0, nc - 1, nc - 1, nc - 1]
        for n in range(1, NL):
            nd = nc << n
            bcl.append(vo.make_bc([[phys[d][0] if self.los[n][d] == 0 else 0, phys[d][1] if self.his[n][d] == nd - 1 else 0] for d in range(3)], 3, 2))
            opd += [0, 0, 0, nd - 1, nd - 1, nd - 1]
        self.obcs = (vo.CBc * NL)(*bcl)
        self.opm = vo.ivec([0, 0, 0])
        self.opd = vo.ivec(opd)
        self.odx = (C.c_double * (3 * NL))(*sum(self.dx, []))
        self._mfs = []

    def ofabs(self, ng, nc, nodal=(0, 0, 0)):
        return [self.vo.Fab(self.los[n], self.his[n], ng, nc, nodal) for n in range(self.nlev)]

    def gmfs(self, ofabs):
        """GPU multifabs (per level) holding the oracle fabs' contents (the fine one cut into the GPU's boxes)"""
        out = []
        for lev, of in enumerate(ofabs):
            mf = self.bl.MultiFab(self.mla, lev, of.nc, of.ng, of.nodal)
            for i in range(mf.nfabs()):
                lo, hi = mf.get_box(i)
                sl = tuple(slice(lo[d] - of.lo[d], hi[d] - of.lo[d] + 1 + of.nodal[d] + 2 * of.ng) for d in range(3))
                mf.from_numpy(np.asfortranarray(of.a[sl]), i)
            self._mfs.append(mf); out.append(mf)
        return out

    def gather(self, mf, of):
        """the GPU level multifab assembled in the shape of the oracle fab (valid + ghost of each box; later boxes win on overlaps
        of ghost regions, valid data always last)"""
        out = np.full(of.a.shape, np.nan, order="F")
        for phase in (0, 1):
            for i in range(mf.nfabs()):
                lo, hi = mf.get_box(i)
                a = mf.to_numpy(i)
                g = of.ng
                if phase == 0:
                    sl = tuple(slice(lo[d] - of.lo[d], hi[d] - of.lo[d] + 1 + of.nodal[d] + 2 * g) for d in range(3))
                    out[sl] = a
                else:
                    sl = tuple(slice(lo[d] - of.lo[d] + g, hi[d] - of.lo[d] + 1 + of.nodal[d] + g) for d in range(3))
                    out[sl] = a[tuple(slice(g, a.shape[d] - g) for d in range(3))] if g else a
        return out

    def smooth(self, of, lev, amp=1.0, base=0.0):
        h = self.dx[lev][0]
        idx = [np.arange(of.lo[d] - of.ng, of.hi[d] + of.nodal[d] + of.ng + 1) for d in range(3)]
        X, Y, Z = np.meshgrid(*[(idx[d] + 0.5) * h for d in range(3)], indexing="ij")
        for c in range(of.nc):
            of.a[..., c] = base + amp * (np.sin(2 * np.pi * (X + 0.3 * c)) * np.cos(2 * np.pi * Y) * np.sin(np.pi * Z + 0.2) + 0.3 * np.cos(4 * np.pi * X * Y + c))

    def close(self):
        for m in self._mfs:
            m.destroy()
        self.bct.destroy(); self.mla.destroy()


def test_transfer_operators_bits(gpu, oracle):
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23))
    L = vo.lib()
    # ml_cc_restriction and fill_ghost_cells / ml_restrict_and_fill on a 2-component ng = 3 state
    s = K.ofabs(3, 2)
    for lev in range(2):
        K.smooth(s[lev], lev, 0.3, 2.0)
    g = K.gmfs(s)
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(s), 0, 3, 2, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    adv.ml_restrict_and_fill(g, 0, 3, 2, K.bct)
    for lev in range(2):
        assert_bits(K.gather(g[lev], s[lev]), s[lev].a, "ml_restrict_and_fill level %d" % lev)
    # ml_edge_restriction + create_umac_grown on a face field
    for d in range(3):
        nd = tuple(1 if t == d else 0 for t in range(3))
        u = K.ofabs(1, 1, nd)
        for lev in range(2):
            K.smooth(u[lev], lev)
        gu = K.gmfs(u)
        L.vo_ml_edge_restriction(u[0].ref, u[1].ref, d)
        adv.ml_edge_restriction(gu[0], gu[1], d)
        assert_bits(K.gather(gu[0], u[0]), u[0].a, "ml_edge_restriction dir %d" % d)
        L.vo_create_umac_grown(u[1].ref, u[0].ref, d)
        adv.create_umac_grown(gu[1], gu[0], d)
        assert_bits(K.gather(gu[1], u[1]), u[1].a, "create_umac_grown dir %d" % d)
    K.close()


def _mac_case(K, vo):
    L = vo.lib()
    NL = K.nlev
    rho = K.ofabs(3, 2)
    for lev in range(NL):
        K.smooth(rho[lev], lev, 0.2, 1.5)
    L.vo_ml_restrict_and_fill(NL, vo.fab_ptr_array(rho), 0, 3, 2, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    um = []
    for lev in range(NL):
        for d in range(3):
            f = K.ofabs(1, 1, tuple(1 if t == d else 0 for t in range(3)))[lev]
            K.smooth(f, lev, 1.0 + 0.1 * d)
            um.append(f)
    for d in range(3):                                    # wall-normal MAC velocity zero on the domain boundary; levels consistent
        sl = [slice(None)] * 4; sl[d] = 1; um[d].a[tuple(sl)] = 0.0; sl[d] = -2; um[d].a[tuple(sl)] = 0.0
        for lev in range(NL - 1, 0, -1):
            L.vo_ml_edge_restriction(um[3 * (lev - 1) + d].ref, um[3 * lev + d].ref, d)
    rhs = K.ofabs(1, 1)
    return rho, um, rhs


def _div(um3, h):
    U, V, W = (m[1:-1, 1:-1, 1:-1, 0] for m in um3)
    return (U[1:, :, :] - U[:-1, :, :]) / h + (V[:, 1:, :] - V[:, :-1, :]) / h + (W[:, :, 1:] - W[:, :, :-1]) / h


@pytest.mark.parametrize("split", [1, 2])
def test_ml_macproject(gpu, oracle, split):
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23), split=split)
    L = vo.lib()
    rho, um, rhs = _mac_case(K, vo)
    grho, grhs = K.gmfs(rho), K.gmfs(rhs)
    gum = [K.gmfs([um[d], um[3 + d]]) for d in range(3)]               # gum[d][lev]
    st = vo.CMgStat()
    L.vo_ml_macproject(2, vo.fab_ptr_array(um), vo.fab_ptr_array(rho), vo.fab_ptr_array(rhs), K.odx, K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.macproject(K.mla, [[gum[d][lev] for d in range(3)] for lev in range(2)], grho, grhs, K.dx, K.bct, 3 + 2 + 1)
    it_gpu = adv.last_solver_stats("mac")[0]
    assert st.cycles < 40 and abs(it_gpu - st.cycles) <= (0 if split == 1 else 1), (it_gpu, st.cycles)
    scale = max(np.abs(m.a).max() for m in um)
    got = [[K.gather(gum[d][lev], um[3 * lev + d]) for d in range(3)] for lev in range(2)]
    for lev in range(2):
        for d in range(3):
            a, b = got[lev][d][1:-1, 1:-1, 1:-1], um[3 * lev + d].a[1:-1, 1:-1, 1:-1]
            assert np.abs(a - b).max() <= 1e-9 * scale, "umac level %d dir %d differs by %.3e" % (lev, d, np.abs(a - b).max())
    # composite divergence of the GPU result
    df, dc = _div(got[1], K.dx[1][0]), _div(got[0], K.dx[0][0])
    tol = 1e-8 * st.res0
    assert np.abs(df).max() <= tol and np.abs(dc).max() <= tol, (np.abs(df).max(), np.abs(dc).max(), st.res0)
    K.close()


@pytest.mark.parametrize("split", [1, 2])
def test_ml_hgproject(gpu, oracle, split):
    """hgproject on two levels (split = 2: the fine level cut into two boxes on the GPU, one box in the oracle): composite nodal solve (interface nodes slaved to the coarse level, Galerkin equations at the coarse
    nodes of the interface), then gradient / velocity / pressure updates and ml_restrict_and_fill(unew)"""
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23), split=split)
    L = vo.lib()
    unew, uold, rhoh, gp, p = K.ofabs(3, 3), K.ofabs(3, 3), K.ofabs(1, 1), K.ofabs(1, 3), K.ofabs(1, 1, (1, 1, 1))
    for lev in range(2):
        K.smooth(unew[lev], lev, 1.0); K.smooth(rhoh[lev], lev, 0.2, 1.5); K.smooth(gp[lev], lev, 0.1)
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(unew), 0, 0, 3, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(rhoh), 0, 3, 1, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(gp), 0, 6, 3, 1, K.obcs, K.opm, K.opd, C.byref(K.prm))
    for lev in range(2):
        uold[lev].a[...] = 0.5 * unew[lev].a
    gun, guo, grh, ggp, gpp = K.gmfs(unew), K.gmfs(uold), K.gmfs(rhoh), K.gmfs(gp), K.gmfs(p)
    dt = 0.01
    st = vo.CMgStat()
    L.vo_ml_hgproject(2, vo.REGULAR_TIMESTEP, vo.fab_ptr_array(unew), vo.fab_ptr_array(uold), vo.fab_ptr_array(rhoh), vo.fab_ptr_array(p), vo.fab_ptr_array(gp),
                      K.odx, C.c_double(dt), K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.hgproject(vo.REGULAR_TIMESTEP, K.mla, gun, guo, grh, gpp, ggp, K.dx, dt, K.bct, 3 + 2 + 1)
    it_gpu = adv.last_solver_stats("hg")[0]
    assert st.cycles < 40 and it_gpu == st.cycles, (it_gpu, st.cycles)
    for lev in range(2):
        a, b = K.gather(gun[lev], unew[lev]), unew[lev].a
        assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), "unew level %d: %.3e" % (lev, np.abs(a - b).max())
        a, b = K.gather(ggp[lev], gp[lev])[1:-1, 1:-1, 1:-1], gp[lev].a[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-8 * np.abs(b).max(), "gp level %d: %.3e" % (lev, np.abs(a - b).max())
        a, b = K.gather(gpp[lev], p[lev])[1:-1, 1:-1, 1:-1], p[lev].a[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-8 * np.abs(b).max(), "p level %d: %.3e" % (lev, np.abs(a - b).max())
    K.close()


@pytest.mark.parametrize("split", [1, 2])
def test_two_level_advance(gpu, oracle, split):
    """BASELINE.json configs[3] in miniature: bubble on a 16^3 base grid with the centre refined (fixed grids), three steps of
    advance_timestep on both levels, HIP vs oracle.  Tolerance 1e-8 relative (two FAC solves per step at 1e-10 / 1e-11)."""
    from varden_amd import advance as adv
    from varden_amd import driver
    vo = oracle
    flo, fhi = (8, 8, 8), (23, 23, 23)
    O = vo.Sim2L(16, flo, fhi, WALLS)
    fboxes = [(flo, fhi)] if split == 1 else [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 23, 23))]
    G = driver.VardenAMR(16, fboxes, WALLS)
    assert G.dt == O.dt
    for _ in range(3):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-12 * O.dt
        assert adv.last_solver_stats("mac")[0] == O.mgstat[0].cycles and adv.last_solver_stats("hg")[0] == O.mgstat[1].cycles
    def level_valid(mfl, n):
        a = np.concatenate([mfl[n].to_numpy(i)[3:-3, 3:-3, 3:-3] for i in range(mfl[n].nfabs())], axis=0)
        return a
    for n in range(2):
        for nm, gm, om in (("u", G.unew, O.unew[n]), ("s", G.snew, O.snew[n])):
            a, b = level_valid(gm, n), om.valid()
            scale = max(np.abs(b).max(), 1e-300)
            assert np.abs(a - b).max() <= 1e-8 * scale, "level %d %s differs by %.3e (scale %.3e)" % (n, nm, np.abs(a - b).max(), scale)
    # the refined bubble stays mirror-symmetric and the coarse level under the fine box is its average
    s1 = level_valid(G.snew, 1)[..., 0]
    assert np.abs(s1 - s1[::-1]).max() <= 1e-9
    s0 = G.snew[0].to_numpy()[3:-3, 3:-3, 3:-3, 0][4:12, 4:12, 4:12]
    avg = s1.reshape(8, 2, 8, 2, 8, 2).mean(axis=(1, 3, 5))
    assert np.abs(s0 - avg).max() <= 1e-13
    G.close()


FINER = [((24, 24, 24), (39, 39, 39))]        # level 2 of the three-level cases: 16^3 cells over the centre of level 1


def test_three_level_macproject(gpu, oracle):
    """the composite MAC solve on three nested levels (precursor of BASELINE.json configs[4]): same FAC iteration count as the
    oracle, MAC velocities to 1e-9, discretely divergence-free on every level"""
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23), finer=FINER)
    L = vo.lib()
    rho, um, rhs = _mac_case(K, vo)
    grho, grhs = K.gmfs(rho), K.gmfs(rhs)
    gum = [K.gmfs([um[3 * lev + d] for lev in range(3)]) for d in range(3)]               # gum[d][lev]
    st = vo.CMgStat()
    L.vo_ml_macproject(3, vo.fab_ptr_array(um), vo.fab_ptr_array(rho), vo.fab_ptr_array(rhs), K.odx, K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.macproject(K.mla, [[gum[d][lev] for d in range(3)] for lev in range(3)], grho, grhs, K.dx, K.bct, 3 + 2 + 1)
    it_gpu = adv.last_solver_stats("mac")[0]
    assert st.cycles < 40 and it_gpu == st.cycles, (it_gpu, st.cycles)
    scale = max(np.abs(m.a).max() for m in um)
    got = [[K.gather(gum[d][lev], um[3 * lev + d]) for d in range(3)] for lev in range(3)]
    for lev in range(3):
        for d in range(3):
            a, b = got[lev][d][1:-1, 1:-1, 1:-1], um[3 * lev + d].a[1:-1, 1:-1, 1:-1]
            assert np.abs(a - b).max() <= 1e-9 * scale, "umac level %d dir %d differs by %.3e" % (lev, d, np.abs(a - b).max())
        assert np.abs(_div(got[lev], K.dx[lev][0])).max() <= 1e-8 * st.res0, lev
    K.close()


def test_three_level_hgproject(gpu, oracle):
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23), finer=FINER)
    L = vo.lib()
    unew, uold, rhoh, gp, p = K.ofabs(3, 3), K.ofabs(3, 3), K.ofabs(1, 1), K.ofabs(1, 3), K.ofabs(1, 1, (1, 1, 1))
    for lev in range(3):
        K.smooth(unew[lev], lev, 1.0); K.smooth(rhoh[lev], lev, 0.2, 1.5); K.smooth(gp[lev], lev, 0.1)
    L.vo_ml_restrict_and_fill(3, vo.fab_ptr_array(unew), 0, 0, 3, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(3, vo.fab_ptr_array(rhoh), 0, 3, 1, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(3, vo.fab_ptr_array(gp), 0, 6, 3, 1, K.obcs, K.opm, K.opd, C.byref(K.prm))
    for lev in range(3):
        uold[lev].a[...] = 0.5 * unew[lev].a
    gun, guo, grh, ggp, gpp = K.gmfs(unew), K.gmfs(uold), K.gmfs(rhoh), K.gmfs(gp), K.gmfs(p)
    dt = 0.01
    st = vo.CMgStat()
    L.vo_ml_hgproject(3, vo.REGULAR_TIMESTEP, vo.fab_ptr_array(unew), vo.fab_ptr_array(uold), vo.fab_ptr_array(rhoh), vo.fab_ptr_array(p), vo.fab_ptr_array(gp),
                      K.odx, C.c_double(dt), K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.hgproject(vo.REGULAR_TIMESTEP, K.mla, gun, guo, grh, gpp, ggp, K.dx, dt, K.bct, 3 + 2 + 1)
    it_gpu = adv.last_solver_stats("hg")[0]
    assert st.cycles < 40 and it_gpu == st.cycles, (it_gpu, st.cycles)
    for lev in range(3):
        a, b = K.gather(gun[lev], unew[lev]), unew[lev].a
        assert np.abs(a - b).max() <= 1e-8 * np.abs(b).max(), "unew level %d: %.3e" % (lev, np.abs(a - b).max())
        a, b = K.gather(ggp[lev], gp[lev])[1:-1, 1:-1, 1:-1], gp[lev].a[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-7 * np.abs(b).max(), "gp level %d: %.3e" % (lev, np.abs(a - b).max())
    K.close()


def test_three_level_advance(gpu, oracle):
    """base 16^3 + two nested refinements over the bubble (fixed grids), three steps of advance_timestep on all three levels, HIP vs
    oracle.  Tolerance 1e-7 relative (composite solves at 1e-10)."""
    from varden_amd import advance as adv
    from varden_amd import driver
    vo = oracle
    flo, fhi = (8, 8, 8), (23, 23, 23)
    O = vo.SimML(16, [(flo, fhi)] + FINER, WALLS)
    G = driver.VardenAMR(16, [(flo, fhi)], WALLS, finer_levels=[FINER])
    assert G.dt == O.dt
    for _ in range(3):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-10 * O.dt
        assert adv.last_solver_stats("mac")[0] == O.mgstat[0].cycles and adv.last_solver_stats("hg")[0] == O.mgstat[1].cycles
    for n in range(3):
        for nm, gm, om in (("u", G.unew, O.unew[n]), ("s", G.snew, O.snew[n])):
            a, b = gm[n].to_numpy(0)[3:-3, 3:-3, 3:-3], om.valid()
            scale = max(np.abs(b).max(), 1e-300)
            assert np.abs(a - b).max() <= 1e-7 * scale, "level %d %s differs by %.3e (scale %.3e)" % (n, nm, np.abs(a - b).max(), scale)
    s2 = G.snew[2].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
    assert np.abs(s2 - s2[::-1]).max() <= 1e-9
    G.close()


# the fine region 8..23 cut so that box 0 shares only PART of a face with boxes 1 and 2; the coarse level cut into four boxes
IRREG_F = [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 15, 23)), ((16, 16, 8), (23, 23, 23))]
QUAD_C = [((0, 0, 0), (7, 7, 15)), ((8, 0, 0), (15, 7, 15)), ((0, 8, 0), (7, 15, 15)), ((8, 8, 0), (15, 15, 15))]


@pytest.mark.parametrize("fb,cb", [(IRREG_F, None), (None, QUAD_C), (IRREG_F, QUAD_C)])
def test_ml_projections_on_irregular_box_unions(gpu, oracle, fb, cb):
    """the composite solves do not depend on how a level is cut into boxes: fine boxes that share partial faces and a multi-box
    coarse level (GPU) against one box per level (oracle): same FAC iteration counts (+-1 for the cell-centred solve whose red-black
    order is per box), velocities to 1e-9"""
    from varden_amd import advance as adv
    vo = oracle
    K = Amr2(16, (8, 8, 8), (23, 23, 23), fboxes=fb, cboxes=cb)
    L = vo.lib()
    # MAC projection
    rho, um, rhs = _mac_case(K, vo)
    grho, grhs = K.gmfs(rho), K.gmfs(rhs)
    gum = [K.gmfs([um[d], um[3 + d]]) for d in range(3)]
    st = vo.CMgStat()
    L.vo_ml_macproject(2, vo.fab_ptr_array(um), vo.fab_ptr_array(rho), vo.fab_ptr_array(rhs), K.odx, K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.macproject(K.mla, [[gum[d][lev] for d in range(3)] for lev in range(2)], grho, grhs, K.dx, K.bct, 3 + 2 + 1)
    assert abs(adv.last_solver_stats("mac")[0] - st.cycles) <= 1, (adv.last_solver_stats("mac")[0], st.cycles)
    scale = max(np.abs(m.a).max() for m in um)
    for lev in range(2):
        for d in range(3):
            a, b = K.gather(gum[d][lev], um[3 * lev + d])[1:-1, 1:-1, 1:-1], um[3 * lev + d].a[1:-1, 1:-1, 1:-1]
            assert np.abs(a - b).max() <= 1e-9 * scale, "umac level %d dir %d differs by %.3e" % (lev, d, np.abs(a - b).max())
    # HG projection
    unew, uold, rhoh, gp, p = K.ofabs(3, 3), K.ofabs(3, 3), K.ofabs(1, 1), K.ofabs(1, 3), K.ofabs(1, 1, (1, 1, 1))
    for lev in range(2):
        K.smooth(unew[lev], lev, 1.0); K.smooth(rhoh[lev], lev, 0.2, 1.5); K.smooth(gp[lev], lev, 0.1)
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(unew), 0, 0, 3, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(rhoh), 0, 3, 1, 0, K.obcs, K.opm, K.opd, C.byref(K.prm))
    L.vo_ml_restrict_and_fill(2, vo.fab_ptr_array(gp), 0, 6, 3, 1, K.obcs, K.opm, K.opd, C.byref(K.prm))
    for lev in range(2):
        uold[lev].a[...] = 0.5 * unew[lev].a
    gun, guo, grh, ggp, gpp = K.gmfs(unew), K.gmfs(uold), K.gmfs(rhoh), K.gmfs(gp), K.gmfs(p)
    L.vo_ml_hgproject(2, vo.REGULAR_TIMESTEP, vo.fab_ptr_array(unew), vo.fab_ptr_array(uold), vo.fab_ptr_array(rhoh), vo.fab_ptr_array(p), vo.fab_ptr_array(gp),
                      K.odx, C.c_double(0.01), K.obcs, K.opm, K.opd, C.byref(K.prm), C.byref(st))
    adv.hgproject(vo.REGULAR_TIMESTEP, K.mla, gun, guo, grh, gpp, ggp, K.dx, 0.01, K.bct, 3 + 2 + 1)
    assert adv.last_solver_stats("hg")[0] == st.cycles, (adv.last_solver_stats("hg")[0], st.cycles)
    for lev in range(2):
        g = 3
        a, b = K.gather(gun[lev], unew[lev])[g:-g, g:-g, g:-g], unew[lev].a[g:-g, g:-g, g:-g]
        assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), "unew level %d: %.3e" % (lev, np.abs(a - b).max())
        a, b = K.gather(gpp[lev], p[lev])[1:-1, 1:-1, 1:-1], p[lev].a[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-8 * np.abs(b).max(), "p level %d: %.3e" % (lev, np.abs(a - b).max())
    K.close()


def _cells(b):
    return int(np.prod([b[1][d] - b[0][d] + 1 for d in range(3)]))


@pytest.mark.parametrize("max_levs", [2, 3])
def test_tagged_grids_properties_and_run(gpu, max_levs):
    """tag_boxes + make_new_grids on the 32^3 bubble (src/initialize.f90:152-342): the boxes of every new level are disjoint,
    blocking-factor aligned, inside the domain, cover every tagged cell (rho > 1.01 / 1.1 of the analytic initial data, grown by
    the buffer), and nest in the level below with two of its cells to spare; the hierarchy then runs advance_timestep: both
    composite solves converge, the coarse level stays mirror-symmetric and equal to the average of the level above it."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    nc = 32
    levels = driver.VardenAMR.tagged_grids(nc, WALLS, default_params(cflfac=0.9), max_levs=max_levs, max_grid_size=32)
    assert len(levels) == max_levs - 1
    cover_prev = None
    for n, lb in enumerate(levels, start=1):
        nd = nc << n
        cover = np.zeros((nd,) * 3, dtype=np.int32)
        for lo, hi in lb:
            assert all(lo[d] % 8 == 0 and (hi[d] + 1) % 8 == 0 and 0 <= lo[d] <= hi[d] < nd and hi[d] - lo[d] + 1 <= 32 for d in range(3)), (lo, hi)
            cover[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] += 1
        assert cover.max() == 1, "boxes of level %d overlap" % n
        # every tagged cell of the parent level (analytic rho, threshold of tag_boxes.f90) is covered
        h = 1.0 / (nd // 2)
        x = (np.arange(nd // 2) + 0.5) * h
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        r = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2)
        rho = 1.0 + 0.5 * (10.0 - 1.0) * (1.0 - np.tanh(30.0 * (r - 0.1)))          # initdata.f90:212-238, densfact 10
        tagged = rho > (1.01 if n == 1 else 1.1)
        if cover_prev is not None:
            tagged &= cover_prev.astype(bool)                 # only cells of the parent level can be tagged
        cov_c = cover.reshape(nd // 2, 2, nd // 2, 2, nd // 2, 2).min(axis=(1, 3, 5)).astype(bool)
        if n == 1:
            assert (cov_c | ~tagged).all(), "a tagged cell of level %d is not covered" % (n - 1)
        else:                                                 # tags outside the nesting region are dropped by construction
            assert (cov_c & tagged).sum() >= 0.9 * tagged.sum()
            # nesting: the parent cells under the new level, grown by 2, lie inside the parent level
            g = cov_c.copy()
            for d in range(3):
                for sft in (1, 2):
                    g |= np.roll(cov_c, sft, axis=d) | np.roll(cov_c, -sft, axis=d)
            assert (cover_prev.astype(bool) | ~g).all(), "level %d is not nested in level %d" % (n, n - 1)
        cover_prev = cover
    G = driver.VardenAMR(nc, levels[0], WALLS, params=default_params(cflfac=0.9), finer_levels=levels[1:])
    for _ in range(2):
        G.step()
        assert adv.last_solver_stats("mac")[0] < 40 and adv.last_solver_stats("hg")[0] < 40
        assert adv.last_solver_stats("mac")[2] <= 1e-10 * adv.last_solver_stats("mac")[1]
    s0 = G.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
    assert np.isfinite(s0).all()
    assert np.abs(s0 - s0[::-1]).max() <= 1e-8 and np.abs(s0 - s0[:, ::-1]).max() <= 1e-8
    # coarse cells under level 1 = mean of their 8 children
    for i, (lo, hi) in enumerate(levels[0]):
        f = G.snew[1].to_numpy(i)[3:-3, 3:-3, 3:-3, 0]
        m = f.reshape(f.shape[0] // 2, 2, f.shape[1] // 2, 2, f.shape[2] // 2, 2).mean(axis=(1, 3, 5))
        c = s0[lo[0] // 2:hi[0] // 2 + 1, lo[1] // 2:hi[1] // 2 + 1, lo[2] // 2:hi[2] // 2 + 1]
        assert np.abs(c - m).max() <= 1e-13
    G.close()


INOUT_BC = [[11, 12], [15, 15], [15, 15]]          # inputs_advect_3d: inflow x-lo, outflow x-hi, no-slip elsewhere


@pytest.mark.parametrize("nc,max_levs,case", [(32, 2, "bubble"), (32, 3, "bubble"), (64, 2, "bubble"), (64, 3, "bubble"), (32, 3, "bubble-viscous"), (32, 3, "advect-viscous"),
                                              (32, 3, "bubble-base-in-eight"), (128, 2, "bubble"), (32, 2, "advect-periodic-x"), (64, 3, "advect-periodic-x")])
def test_tagged_hierarchy_against_the_box_list_oracle(gpu, oracle, nc, max_levs, case):
    """BASELINE.json configs[3] / [4] in small: the refined levels are the boxes make_new_grids returns for the tagged bubble (tag_boxes.f90:65-94: rho > 1.01 /
    rho > 1.1) -- unions that are not rectangles, re-entrant interface edges, boxes of a few cells -- and the ORACLE RUNS THE SAME BOX LISTS (oracle/vo.h:
    level arrays with a cell mask, MAC velocities and the Godunov kernels box by box; VERDICT r4 missing 3).  Start-up (initial projection + one pressure
    iteration) and two steps: dt bit for bit, the FAC iteration counts of both composite solves equal in every call, u / rho / tracer to 1e-9 on every box
    of every level, the pressure to 1e-6; and the composite mass is conserved to round-off (the conservative fluxes are restricted, mkflux.f90:137-146).
    Bases of 32^3, 64^3 and 128^3 cells (the last: half the linear size of configs[3], two levels, one step; round 6: configs[3] and configs[4] THEMSELVES are held against
    oracle-written fixtures in tests/test_fullsize_gpu.py::test_tagged_hierarchy_step_at_256, which replaced the 80-second three-level run at a 128^3 base here).
    Cases: the inviscid bubble between walls (the bench's configuration); the same with visc_coef = 0.001 as exec/test/inputs_bubble_3d and inputs_3d-regt
    have it (explicit diffusive term + composite Crank-Nicolson solves per velocity component); the advected blob of inputs_advect_3d (prob_type 2, inflow /
    outflow: Dirichlet sides in both composite solves, inhomogeneous boundary data in the viscous ones); round 6, a PERIODIC hierarchy against the oracle: the same
    blob carried at u = 1 THROUGH periodic x faces while gravity pulls it down (slip walls in y, no-slip in z) -- level 0 wraps in every operator of both composite
    solves, the Godunov ghost cells and mkumac; the refined levels follow the blob in the interior (the oracle takes periodic faces on level 0 only,
    oracle/vo_amr.c: require_periodic_ok; a face treated as a wall would stop a unit through-flow)."""
    from tests.util import params_for
    from varden_amd import advance as adv
    from varden_amd import driver
    vo = oracle
    phys, prob, grav, kw = WALLS, 1, -9.8, dict(cflfac=0.9)
    if case == "bubble-viscous":
        kw.update(visc_coef=0.001)
    elif case == "advect-viscous":
        phys, prob, grav = INOUT_BC, 2, 0.0
        kw.update(visc_coef=0.001)
    elif case == "advect-periodic-x":
        phys, prob, grav = [[-1, -1], [14, 14], [15, 15]], 2, -9.8
    base = None
    if case == "bubble-base-in-eight":                      # level 0 cut into 2 x 2 x 2 boxes, as `bench.py --config amr3` cuts it for several ranks (configs[4])
        hb = nc // 2
        base = [((i * hb, j * hb, k * hb), ((i + 1) * hb - 1, (j + 1) * hb - 1, (k + 1) * hb - 1)) for k in range(2) for j in range(2) for i in range(2)]
    levels = driver.VardenAMR.tagged_grids(nc, phys, params_for(phys, **kw), prob_type=prob, max_levs=max_levs, max_grid_size=32 if prob == 1 else 16, base_boxes=base)
    assert len(levels) == max_levs - 1 and len(levels[0]) > 1, "the tagged blob should give unions of several boxes: %r" % ([len(lb) for lb in levels],)
    G = driver.VardenAMR(nc, levels[0], phys, params=params_for(phys, **kw), prob_type=prob, grav=grav, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1,
                         base_boxes=base)
    O = vo.SimML(nc, levels, phys, prm=params_for(phys, **kw), prob_type=prob, grav=grav, init_shrink=0.1, init_iter=1, do_initial_projection=1, base_boxes=base)
    assert G.initial_projection_stat[0] == O.initial_projection_stat[0], "initial projection: FAC iterations %r (GPU) vs %r (oracle)" % (G.initial_projection_stat[0], O.initial_projection_stat[0])
    assert G.dt == O.dt

    def mass():
        m = 0.0
        for n in range(O.nlev):
            msk = O.levels[n].mask()
            if n + 1 < O.nlev:
                f = O.levels[n + 1]
                fm = f.mask()[::2, ::2, ::2]
                o = [f.lo[d] // 2 - O.levels[n].lo[d] for d in range(3)]
                cov = np.zeros_like(msk)
                cov[o[0]:o[0] + fm.shape[0], o[1]:o[1] + fm.shape[1], o[2]:o[2] + fm.shape[2]] = fm
                msk = msk & ~cov
            for i in range(G.sold[n].nfabs()):
                lo, hi = G.sold[n].get_box(i)
                sl = tuple(slice(lo[d] - O.levels[n].lo[d], hi[d] - O.levels[n].lo[d] + 1) for d in range(3))
                m += (G.sold[n].to_numpy(i)[3:-3, 3:-3, 3:-3, 0] * msk[sl]).sum() / 8.0 ** n
        return m
    m0 = mass()
    for step in range(1 if nc == 128 else 2):                 # (128^3 base: half the linear size of configs[3], one step)
        O.step(); G.step()
        assert G.dt == O.dt, "dt diverged at step %d: %r vs %r" % (step, G.dt, O.dt)
        cg = (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
        co = (O.mgstat[0].cycles, O.mgstat[1].cycles)
        assert cg == co, "base %d^3, %d levels, step %d: FAC iterations (MAC, HG) %r on the GPU, %r in the oracle" % (nc, max_levs, step, cg, co)
        for n in range(O.nlev):
            olo = O.levels[n].lo
            for nm, gm, om, g, tol in (("u", G.uold[n], O.uold[n], 3, 1e-9), ("s", G.sold[n], O.sold[n], 3, 1e-9), ("gp", G.gp[n], O.gp[n], 1, 1e-6)):
                scale = max(float(np.abs(om.valid()).max()), 1e-300)
                for i in range(gm.nfabs()):
                    lo, hi = gm.get_box(i)
                    a = gm.to_numpy(i)[g:-g, g:-g, g:-g]
                    b = om.valid()[tuple(slice(lo[d] - olo[d], hi[d] - olo[d] + 1) for d in range(3))]
                    err = float(np.abs(a - b).max())
                    assert err <= tol * scale, "level %d box %d step %d: %s differs by %.3e (scale %.3e)" % (n, i, step, nm, err, scale)
    m1 = mass()
    if phys is WALLS or case == "advect-periodic-x":        # (inflow / outflow: mass enters and leaves)
        assert abs(m1 - m0) <= 1e-12 * m0, "composite mass drifted by %.3e" % ((m1 - m0) / m0)
    G.close()


def test_regridding_keeps_its_memory_and_its_addresses(gpu):
    """round 6: the arena of per-step temporaries is an address range reserved once with physical chunks mapped on demand, the fields of a hierarchy sit on pooled chunks
    (runtime.hip: arena_map_to, field_alloc) -- until then a regrid whose layout wanted a larger arena freed and allocated 120 GB (3-7 s) and every regrid returned the
    fields' memory to the driver.  A 128^3 base with three levels, viscous, regrid_int = 2, twelve steps (six regrids, the field sizes change every time): the memory
    backing the arena stops growing after the first steps, the device memory in use settles (no chunk leaks from the pool, no plan or descriptor set outlives its layout)
    and the run stays finite and converged."""
    import ctypes as C
    import torch
    from varden_amd import advance as adv, capi, driver
    from varden_amd.capi import default_params
    prm = default_params(cflfac=0.9, visc_coef=0.001)
    nc = 128
    levels = driver.VardenAMR.tagged_grids(nc, WALLS, prm, max_levs=3, max_grid_size=64)
    G = driver.VardenAMR(nc, levels[0], WALLS, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, regrid_int=2, max_levs=3, max_grid_size=64)
    hist = []
    for s in range(12):
        G.step()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        rb, pk = C.c_size_t(), C.c_size_t()
        capi.load().vdn_arena_stats(C.byref(rb), C.byref(pk))
        hist.append((total - free, rb.value, pk.value))
        assert adv.last_solver_stats("mac")[2] <= 1e-10 * adv.last_solver_stats("mac")[1]
    assert G.nregrids == 6
    assert hist[-1][1] == hist[3][1], "the arena kept growing: %r" % [h[1] >> 20 for h in hist]
    assert hist[-1][1] - hist[-1][2] <= (1 << 30), "more than one chunk mapped beyond the high-water mark"
    assert hist[-1][0] <= hist[5][0] + (256 << 20), "device memory in use grew over the last six steps (three regrids): %r MB" % [h[0] >> 20 for h in hist]
    assert all(np.isfinite(G.uold[n].to_numpy(0)).all() for n in range(G.nlev))
    G.close()


def test_regridding_run_against_the_box_list_oracle(gpu, oracle):
    """the time loop WITH regridding (src/varden.f90:256-264, src/regrid.f90: tag_boxes + make_new_grids every regrid_int steps, fillpatch,
    ml_nodal_prolongation, copies between the old and the new box lists) on both sides: three levels on a 32^3 base, regrid_int = 2, six steps (three
    regrids on the moving bubble).  The oracle takes the box lists the GPU's make_new_grids returns and moves ITS OWN state onto them
    (voracle.SimML.regrid); from then on the two runs must go on agreeing: dt bit for bit, equal FAC counts, u / rho to 1e-9 on every box."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    vo = oracle
    nc = 32
    levels = driver.VardenAMR.tagged_grids(nc, WALLS, default_params(cflfac=0.9), max_levs=3, max_grid_size=32)
    G = driver.VardenAMR(nc, levels[0], WALLS, params=default_params(cflfac=0.9), finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1,
                         regrid_int=2, max_levs=3, max_grid_size=32)
    O = vo.SimML(nc, levels, WALLS, prm=default_params(cflfac=0.9), init_shrink=0.1, init_iter=1, do_initial_projection=1)
    nreg = 0
    for step in range(6):
        G.step()
        if G.nregrids != nreg:                              # the GPU regridded at the top of this step: the oracle follows with the same grids
            nreg = G.nregrids
            O.regrid([[(tuple(b[0]), tuple(b[1])) for b in lb] for lb in G.boxes[1:]])
        O.step()
        assert G.dt == O.dt, "dt diverged at step %d: %r vs %r" % (step, G.dt, O.dt)
        cg = (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
        co = (O.mgstat[0].cycles, O.mgstat[1].cycles)
        assert cg == co, "step %d: FAC iterations (MAC, HG) %r on the GPU, %r in the oracle" % (step, cg, co)
        assert G.nlev == O.nlev
        for n in range(O.nlev):
            olo = O.levels[n].lo
            for nm, gm, om, g in (("u", G.uold[n], O.uold[n], 3), ("s", G.sold[n], O.sold[n], 3)):
                scale = max(float(np.abs(om.valid()).max()), 1e-300)
                for i in range(gm.nfabs()):
                    lo, hi = gm.get_box(i)
                    a = gm.to_numpy(i)[g:-g, g:-g, g:-g]
                    b = om.valid()[tuple(slice(lo[d] - olo[d], hi[d] - olo[d] + 1) for d in range(3))]
                    err = float(np.abs(a - b).max())
                    assert err <= 1e-9 * scale, "level %d box %d step %d: %s differs by %.3e (scale %.3e)" % (n, i, step, nm, err, scale)
    assert nreg == 3, nreg
    G.close()


def test_fillpatch_and_nodal_prolongation_reproduce_linear_fields(gpu):
    """regrid.f90:311-327: fillpatch (limited linear interpolation) is exact for a linear cell field, ml_nodal_prolongation (trilinear)
    for a trilinear nodal field; copy between layouts moves exactly the points valid in both"""
    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    from varden_amd.capi import default_params
    bl.initialize(default_params(), 0, 1, 0)
    nc = 16
    pd = [((0, 0, 0), (nc - 1,) * 3), ((0, 0, 0), (2 * nc - 1,) * 3)]
    cboxes = [((0, 0, 0), (7, 15, 15)), ((8, 0, 0), (15, 15, 15))]
    fboxes = [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 15, 23)), ((16, 16, 8), (23, 23, 23))]
    mla = bl.MLLayout(pd, [cboxes, fboxes], rr=[(2, 2, 2)])
    lin = lambda x, y, z: 1.0 + 2.0 * x - 3.0 * y + 0.5 * z                                    # noqa: E731
    tri = lambda x, y, z: 1.0 + x - 2.0 * y + 3.0 * z + x * y - 0.5 * y * z + 2.0 * x * y * z     # noqa: E731
    crse, fine = bl.MultiFab(mla, 0, 1, 1), bl.MultiFab(mla, 1, 1, 0)
    pc, pf = bl.MultiFab(mla, 0, 1, 1, (1, 1, 1)), bl.MultiFab(mla, 1, 1, 1, (1, 1, 1))
    for i, (lo, hi) in enumerate(cboxes):
        ax = [(np.arange(lo[d] - 1, hi[d] + 2) + 0.5) / nc for d in range(3)]
        crse.from_numpy(lin(*np.meshgrid(*ax, indexing="ij"))[..., None], i)
        an = [np.arange(lo[d] - 1, hi[d] + 3) / nc for d in range(3)]
        pc.from_numpy(tri(*np.meshgrid(*an, indexing="ij"))[..., None], i)
    adv.fillpatch(fine, crse, 0, 1)
    adv.ml_nodal_prolongation(pf, pc)
    for i, (lo, hi) in enumerate(fboxes):
        ax = [(np.arange(lo[d], hi[d] + 1) + 0.5) / (2 * nc) for d in range(3)]
        assert np.abs(fine.to_numpy(i)[..., 0] - lin(*np.meshgrid(*ax, indexing="ij"))).max() <= 1e-14
        an = [np.arange(lo[d], hi[d] + 2) / (2 * nc) for d in range(3)]
        assert np.abs(pf.to_numpy(i)[1:-1, 1:-1, 1:-1, 0] - tri(*np.meshgrid(*an, indexing="ij"))).max() <= 1e-14
    # copy between two box lists of level 1
    mlb = bl.MLLayout(pd, [cboxes, [((12, 12, 12), (19, 19, 19))]], rr=[(2, 2, 2)])
    other = bl.MultiFab(mlb, 1, 1, 0)
    other.setval(-7.0, all=True)
    adv.copy_layouts(other, 0, fine, 0, 1)
    ax = [(np.arange(12, 20) + 0.5) / (2 * nc) for d in range(3)]
    assert np.abs(other.to_numpy(0)[..., 0] - lin(*np.meshgrid(*ax, indexing="ij"))).max() <= 1e-14
    for m in (crse, fine, pc, pf, other):
        m.destroy()
    mla.destroy(); mlb.destroy()


@pytest.mark.parametrize("split", [1, 2])
def test_regrid_operators_against_the_oracle(gpu, oracle, split):
    """SURVEY.md section 8(f-3), VERDICT r1 item 6: the three per-cell operators of regridding against their CPU restatements, bit for bit.
    tag_boxes_3d (src/tag_boxes.f90:128-216, in the reference tree) for levels 1, 2, 3 and prob_type 1 and 3; fillpatch and
    ml_nodal_prolongation as src/regrid.f90:311-327 calls them (FBoxLib routines: the definitions are ours, stated in oracle/vo_amr.c,
    so agreement shows the device code implements THAT definition).  The clustering of make_new_grids (Berger-Rigoutsos on the host,
    also FBoxLib) has no oracle and stays property-tested: test_tagged_grids_properties_and_run, test_regrid."""
    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    from varden_amd.capi import default_params
    vo, L = oracle, oracle.lib()
    K = Amr2(16, (8, 8, 8), (23, 23, 23), split=split, seed=5)
    # --- tag_boxes: a density field that crosses every threshold (1.01, 1.1, 1.5; 1.2 .. 1.8), and values exactly ON the thresholds
    s = K.ofabs(3, 2)
    for lev in range(2):
        K.smooth(s[lev], lev, 0.45, 1.35)
        v = s[lev].a[3:-3, 3:-3, 3:-3, 0]
        v[0, 0, :6] = [1.01, 1.1, 1.5, 1.2, 1.8, np.nextafter(1.01, 2.0)]          # '>' is strict (tag_boxes.f90:147)
    g = K.gmfs(s)
    for prob_type in (1, 3):
        prm = default_params(); prm.prob_type = prob_type
        bl.initialize(prm, 0, 1, 0)
        for lev in (1, 2, 3):
            for n in range(2):
                of = s[n]
                shape = tuple(of.hi[d] - of.lo[d] + 1 for d in range(3))
                ot = np.zeros(shape, dtype=np.uint8, order="F")
                assert L.vo_tag_boxes(of.ref, lev, prob_type, ot.ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
                gt = adv.tag_boxes(g[n], lev)                                       # over the level's domain
                sl = tuple(slice(of.lo[d], of.hi[d] + 1) for d in range(3))
                assert np.array_equal(gt[sl], ot), "tags differ: prob_type %d lev %d level-index %d" % (prob_type, lev, n)
                assert gt.sum() == ot.sum() and 0 < ot.sum() < ot.size              # nothing tagged outside the level's boxes
    ot = np.zeros(1, dtype=np.uint8)
    assert L.vo_tag_boxes(s[0].ref, 1, 4, ot.ctypes.data_as(C.POINTER(C.c_ubyte))) == -1    # bl_error('Unsupported prob_type'), :212
    bl.initialize(default_params(), 0, 1, 0)
    # --- fillpatch: the new fine level from the coarse one (coarse ghost cells filled)
    c = K.ofabs(3, 2)
    K.smooth(c[0], 0, 0.7, 1.0)
    c[1].a[...] = -99.0
    gc = K.gmfs(c)
    L.vo_fillpatch(c[1].ref, c[0].ref, 0, 2)
    adv.fillpatch(gc[1], gc[0], 0, 2)
    assert_bits(K.gather(gc[1], c[1])[3:-3, 3:-3, 3:-3], c[1].a[3:-3, 3:-3, 3:-3], "fillpatch")
    assert c[1].a[3:-3, 3:-3, 3:-3].min() > -50.0                                   # every valid fine cell was written
    # --- ml_nodal_prolongation of the nodal pressure
    p = K.ofabs(1, 1, (1, 1, 1))
    K.smooth(p[0], 0, 1.0, 0.2)
    p[1].a[...] = -99.0
    gp = K.gmfs(p)
    L.vo_nodal_prolongation(p[1].ref, p[0].ref)
    adv.ml_nodal_prolongation(gp[1], gp[0])
    assert_bits(K.gather(gp[1], p[1])[1:-1, 1:-1, 1:-1], p[1].a[1:-1, 1:-1, 1:-1], "ml_nodal_prolongation")
    assert p[1].a[1:-1, 1:-1, 1:-1].min() > -50.0
    K.close()


def test_regrid(gpu):
    """src/regrid.f90: (1) regridding an unchanged state gives the same boxes and, bit for bit, the same data (everything is copied
    from the old level); (2) a hierarchy that starts with ONE large fine box regrids onto the tagged boxes inside it, keeps the old
    fine data there bit for bit, and goes on stepping with the regrid interval of the reference's inputs (regrid_int = 2)."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    nc = 32
    levels = driver.VardenAMR.tagged_grids(nc, WALLS, default_params(cflfac=0.9), max_levs=2, max_grid_size=32)
    G = driver.VardenAMR(nc, levels[0], WALLS, params=default_params(cflfac=0.9), regrid_int=2, max_levs=2, max_grid_size=32)
    G.step()
    before = {k: [[getattr(G, k)[n].to_numpy(i) for i in range(getattr(G, k)[n].nfabs())] for n in range(G.nlev)] for k in ("uold", "sold", "gp", "p")}
    boxes_before = [list(b) for b in G.boxes]
    G.fill_state_ghosts()
    ghosts = {k: [[getattr(G, k)[n].to_numpy(i) for i in range(getattr(G, k)[n].nfabs())] for n in range(G.nlev)] for k in ("uold", "sold", "gp")}
    G.regrid(buf_wid=2)
    assert [sorted(b) for b in G.boxes] == [sorted(b) for b in boxes_before]
    for k in ("uold", "sold", "gp"):
        for n in range(G.nlev):
            for i, bx in enumerate(G.boxes[n]):
                j = boxes_before[n].index(bx)
                assert np.array_equal(getattr(G, k)[n].to_numpy(i), ghosts[k][n][j]), (k, n, i)
    for n in range(G.nlev):
        for i, bx in enumerate(G.boxes[n]):
            j = boxes_before[n].index(bx)
            assert np.array_equal(G.p[n].to_numpy(i)[1:-1, 1:-1, 1:-1], before["p"][n][j][1:-1, 1:-1, 1:-1])
    for _ in range(4):                                       # steps 2..5: regrids before steps 3 and 5
        G.step()
    assert G.nregrids >= 3 and adv.last_solver_stats("hg")[0] < 40
    s0 = G.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
    assert np.isfinite(s0).all() and np.abs(s0 - s0[::-1]).max() <= 1e-8
    G.close()
    # (2) one big fine box -> tagged boxes inside it
    # (regrid_int is switched on after the first step: like the reference, src/varden.f90:256, the driver regrids before step 1 too)
    H = driver.VardenAMR(nc, [((8, 8, 8), (55, 55, 55))], WALLS, params=default_params(cflfac=0.9), regrid_int=-1, max_levs=2, max_grid_size=32)
    H.step()
    big = H.sold[1].to_numpy(0)
    H.regrid_int, H.amr_buf_width = 2, 2
    H.regrid(buf_wid=2)
    assert len(H.boxes[1]) > 1
    for i, (lo, hi) in enumerate(H.boxes[1]):
        assert all(8 <= lo[d] and hi[d] <= 55 for d in range(3))
        a = H.sold[1].to_numpy(i)[3:-3, 3:-3, 3:-3]
        b = big[3 + lo[0] - 8:3 + hi[0] - 8 + 1, 3 + lo[1] - 8:3 + hi[1] - 8 + 1, 3 + lo[2] - 8:3 + hi[2] - 8 + 1]
        assert np.array_equal(a, b)
    H.step(); H.step()
    assert np.isfinite(H.snew[0].to_numpy(0)).all()
    H.close()


@pytest.mark.parametrize("split,nlev", [(1, 2), (2, 2), (1, 3)])
def test_multi_level_viscous_advance(gpu, oracle, split, nlev):
    """exec/test/inputs_bubble_3d in miniature: visc_coef = 0.001 on a refined hierarchy -- the explicit diffusive term per level
    (averaged down), the composite solves of (rho - div mu grad) u = rhs per velocity component (viscsolve.f90:19-306) with the
    no-slip wall values in the ghost cells; HIP vs oracle after three steps, 1e-8 relative"""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    vo = oracle
    flo, fhi = (8, 8, 8), (23, 23, 23)
    finer = FINER if nlev == 3 else []
    O = vo.SimML(16, [(flo, fhi)] + finer, WALLS, prm=default_params(cflfac=0.9, visc_coef=0.001))
    fboxes = [(flo, fhi)] if split == 1 else [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 23, 23))]
    G = driver.VardenAMR(16, fboxes, WALLS, params=default_params(cflfac=0.9, visc_coef=0.001), finer_levels=[finer] if finer else [])
    assert G.dt == O.dt
    for _ in range(3):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-10 * O.dt
        assert adv.last_solver_stats("mac")[0] == O.mgstat[0].cycles and adv.last_solver_stats("hg")[0] == O.mgstat[1].cycles
    for n in range(nlev):
        for nm, gm, om in (("u", G.unew, O.unew[n]), ("s", G.snew, O.snew[n])):
            a = np.concatenate([gm[n].to_numpy(i)[3:-3, 3:-3, 3:-3] for i in range(gm[n].nfabs())], axis=0)
            b = om.valid()
            scale = max(np.abs(b).max(), 1e-300)
            assert np.abs(a - b).max() <= 1e-8 * scale, "level %d %s differs by %.3e (scale %.3e)" % (n, nm, np.abs(a - b).max(), scale)
    G.close()


def test_two_level_start_up_sequence(gpu, oracle):
    """the start-up of src/varden.f90 on two levels: initial projection (hgproject with proj_type = initial_projection, rhohalf = 1,
    dt = 1; varden.f90:126-138), p = gp = 0, first dt, one pressure iteration (advance_timestep with proj_type = pressure_iters,
    varden.f90:460-490), then two regular viscous steps -- HIP vs oracle"""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    vo = oracle
    flo, fhi = (8, 8, 8), (23, 23, 23)
    kw = dict(init_iter=1, do_initial_projection=1)
    O = vo.SimML(16, [(flo, fhi)], WALLS, prm=default_params(cflfac=0.9, visc_coef=0.001), **kw)
    G = driver.VardenAMR(16, [(flo, fhi)], WALLS, params=default_params(cflfac=0.9, visc_coef=0.001), **kw)
    assert G.dt == O.dt
    for n in range(2):
        a, b = G.p[n].to_numpy(0)[1:-1, 1:-1, 1:-1], O.p[n].a[1:-1, 1:-1, 1:-1]
        assert np.abs(a - b).max() <= 1e-7 * max(np.abs(b).max(), 1e-300), "p after the pressure iteration, level %d: %.3e / %.3e" % (n, np.abs(a - b).max(), np.abs(b).max())
    for _ in range(2):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-10 * O.dt
    for n in range(2):
        for nm, gm, om in (("u", G.unew, O.unew[n]), ("s", G.snew, O.snew[n])):
            a, b = gm[n].to_numpy(0)[3:-3, 3:-3, 3:-3], om.valid()
            assert np.abs(a - b).max() <= 1e-8 * max(np.abs(b).max(), 1e-300), "level %d %s differs by %.3e" % (n, nm, np.abs(a - b).max())
    G.close()


def test_periodic_hierarchy_is_translation_invariant(gpu):
    """a hierarchy on a domain that is periodic in x: shifting the whole problem by half a period -- the bubble then sits ON the periodic
    boundary and its fine level is two boxes, one at each end of the domain, talking to each other and to the coarse level through
    periodic images -- must give the shifted solution.  Three steps (MAC, viscous and nodal composite solves each); 1e-8 relative:
    the two runs cut the fine level differently, so only the solver tolerances separate them."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    nc = 16
    phys = [[-1, -1], [15, 15], [15, 15]]

    def bubble(xc):
        def fn(lev, blo, nb, dx):
            g = 3
            ax = [(np.arange(blo[d] - g, blo[d] + nb[d] + g) + 0.5) * dx[d] for d in range(3)]
            X, Y, Z = np.meshgrid(*ax, indexing="ij")
            dxp = (X - xc + 0.5) % 1.0 - 0.5                   # periodic distance in x
            r = np.sqrt(dxp ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2)
            s = np.zeros(X.shape + (2,), order="F")
            s[..., 0] = 1.0 + 0.5 * (10.0 - 1.0) * (1.0 - np.tanh(30.0 * (r - 0.1)))
            s[..., 1] = s[..., 0]
            return np.zeros(X.shape + (3,), order="F"), s
        return fn

    prm = lambda: default_params(cflfac=0.9, visc_coef=0.001)   # noqa: E731
    A = driver.VardenAMR(nc, [((8, 8, 8), (23, 23, 23))], phys, params=prm(), init_fn=bubble(0.5), init_iter=1, do_initial_projection=1)
    for _ in range(3):
        A.step()
    ua0, sa0 = A.unew[0].to_numpy(0)[3:-3, 3:-3, 3:-3], A.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3]
    ua1, sa1 = A.unew[1].to_numpy(0)[3:-3, 3:-3, 3:-3], A.snew[1].to_numpy(0)[3:-3, 3:-3, 3:-3]
    dta, ita = A.dt, (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
    A.close()
    B = driver.VardenAMR(nc, [((24, 8, 8), (31, 23, 23)), ((0, 8, 8), (7, 23, 23))], phys, params=prm(), init_fn=bubble(0.0), init_iter=1, do_initial_projection=1)
    for _ in range(3):
        B.step()
    assert abs(B.dt - dta) <= 1e-9 * dta
    assert abs(adv.last_solver_stats("mac")[0] - ita[0]) <= 1 and abs(adv.last_solver_stats("hg")[0] - ita[1]) <= 1
    ub0, sb0 = B.unew[0].to_numpy(0)[3:-3, 3:-3, 3:-3], B.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3]
    for a, b, nm in ((ua0, ub0, "u"), (sa0, sb0, "s")):
        assert np.abs(np.roll(a, nc // 2, axis=0) - b).max() <= 1e-8 * np.abs(a).max(), "level 0 %s: %.3e" % (nm, np.abs(np.roll(a, nc // 2, axis=0) - b).max())
    # level 1: run A's box 8..23 in x is run B's boxes 24..31 (its low half) and 0..7 (its high half)
    for a, mf, nm in ((ua1, B.unew[1], "u"), (sa1, B.snew[1], "s")):
        lo_half, hi_half = mf.to_numpy(0)[3:-3, 3:-3, 3:-3], mf.to_numpy(1)[3:-3, 3:-3, 3:-3]
        assert np.abs(a[:8] - lo_half).max() <= 1e-8 * np.abs(a).max() and np.abs(a[8:] - hi_half).max() <= 1e-8 * np.abs(a).max(), nm
    assert np.abs(ua1[..., 2]).max() > 0
    B.close()


@pytest.mark.parametrize("nlev", [2, 3])
def test_refined_boxes_across_a_periodic_face_against_the_oracle(gpu, oracle, nlev):
    """round 6: the box-list oracle wraps level 0 but keeps no periodic IMAGES of refined boxes (oracle/vo_amr.c: require_periodic_ok).  What it cannot run directly it checks
    through the translation: the oracle runs the bubble in the MIDDLE of a domain periodic in x (its fine box in the interior); the library runs the same problem shifted by
    half a period -- the bubble ON the periodic face, its fine level two boxes, one at each end of the domain, talking to each other and to level 0 through periodic images in
    every operator of both composite solves, the viscous ones, the ghost fills and the Godunov stencils -- and must give the oracle's fields, shifted.  Three steps, 1e-8 of the
    field's max (the two runs cut the fine level differently: the solver tolerances separate them), dt to 1e-9, FAC counts within one.  nlev = 3: a second refined level, also
    cut in two by the periodic face (the corrections prolonged linearly into level 2 read their coarse neighbours through periodic images of another box)."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    vo = oracle
    nc = 16
    phys = [[-1, -1], [15, 15], [15, 15]]

    def bubble(xc):
        def fn(lev, blo, nb, dx):
            g = 3
            ax = [(np.arange(blo[d] - g, blo[d] + nb[d] + g) + 0.5) * dx[d] for d in range(3)]
            X, Y, Z = np.meshgrid(*ax, indexing="ij")
            dxp = (X - xc + 0.5) % 1.0 - 0.5                   # periodic distance in x
            r = np.sqrt(dxp ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2)
            s = np.zeros(X.shape + (2,), order="F")
            s[..., 0] = 1.0 + 0.5 * (10.0 - 1.0) * (1.0 - np.tanh(30.0 * (r - 0.1)))
            s[..., 1] = s[..., 0]
            return np.zeros(X.shape + (3,), order="F"), s
        return fn

    prm = lambda: default_params(cflfac=0.9, visc_coef=0.001)   # noqa: E731
    obox = [((8, 8, 8), (23, 23, 23))] + ([((24, 24, 24), (39, 39, 39))] if nlev == 3 else [])
    O = vo.SimML(nc, obox, phys, prm=prm(), init_fn=bubble(0.5), init_iter=1, do_initial_projection=1)
    B = driver.VardenAMR(nc, [((24, 8, 8), (31, 23, 23)), ((0, 8, 8), (7, 23, 23))], phys, params=prm(), init_fn=bubble(0.0), init_iter=1, do_initial_projection=1,
                         finer_levels=[[((56, 24, 24), (63, 39, 39)), ((0, 24, 24), (7, 39, 39))]] if nlev == 3 else ())
    assert abs(B.dt - O.dt) <= 1e-9 * O.dt
    for _ in range(3):
        O.step(); B.step()
        assert abs(B.dt - O.dt) <= 1e-9 * O.dt, (B.dt, O.dt)
        assert abs(adv.last_solver_stats("mac")[0] - O.mgstat[0].cycles) <= 1 and abs(adv.last_solver_stats("hg")[0] - O.mgstat[1].cycles) <= 1
    for nm, gm, om in (("u", B.unew, O.unew), ("s", B.snew, O.snew)):
        a0, b0 = om[0].valid(), gm[0].to_numpy(0)[3:-3, 3:-3, 3:-3]
        assert np.abs(np.roll(a0, nc // 2, axis=0) - b0).max() <= 1e-8 * np.abs(a0).max(), "level 0 %s: %.3e" % (nm, np.abs(np.roll(a0, nc // 2, axis=0) - b0).max())
        for n in range(1, nlev):                                # the oracle's box (8..23 / 24..39 in x) is the library's two boxes: its low half at the high end of the domain, its high half at the low end
            a1 = om[n].valid()
            h = a1.shape[0] // 2
            lo_half, hi_half = gm[n].to_numpy(0)[3:-3, 3:-3, 3:-3], gm[n].to_numpy(1)[3:-3, 3:-3, 3:-3]
            assert np.abs(a1[:h] - lo_half).max() <= 1e-8 * np.abs(a1).max() and np.abs(a1[h:] - hi_half).max() <= 1e-8 * np.abs(a1).max(), (nm, n)
    assert np.abs(O.unew[1].valid()[..., 2]).max() > 0
    B.close()


def test_two_level_scalar_diffusion(gpu, oracle):
    """diff_coef > 0 on a refined hierarchy: the explicit diffusive term of the tracer per level (averaged down) and the composite solve
    of (1 - div mu grad) s = rhs (viscsolve.f90:308-515); with visc_coef > 0 as well.  HIP vs oracle after three steps, 1e-8 relative"""
    from varden_amd import driver
    from varden_amd.capi import default_params
    vo = oracle
    flo, fhi = (8, 8, 8), (23, 23, 23)
    kw = dict(cflfac=0.9, visc_coef=0.001, diff_coef=0.002)
    O = vo.SimML(16, [(flo, fhi)], WALLS, prm=default_params(**kw))
    G = driver.VardenAMR(16, [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 23, 23))], WALLS, params=default_params(**kw))
    for _ in range(3):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-10 * O.dt
    for n in range(2):
        for nm, gm, om in (("u", G.unew, O.unew[n]), ("s", G.snew, O.snew[n])):
            a = np.concatenate([gm[n].to_numpy(i)[3:-3, 3:-3, 3:-3] for i in range(gm[n].nfabs())], axis=0)
            b = om.valid()
            assert np.abs(a - b).max() <= 1e-8 * max(np.abs(b).max(), 1e-300), "level %d %s differs by %.3e" % (n, nm, np.abs(a - b).max())
    # the tracer (component 1) has diffused away from the density it started equal to
    assert np.abs(O.snew[1].valid()[..., 1] - O.snew[1].valid()[..., 0]).max() > 1e-6
    G.close()


def test_composite_launch_variants_agree_bit_for_bit(gpu):
    """the launch-level forms of the composite solves change no value: a tagged three-level hierarchy (base 32^3, unions of boxes) run (a) with
    the defaults -- eight planes per workgroup for the light box-batched kernels (vdn_dev.h batch_ppw), the interface interpolation of the
    nodal solve over box faces only, the composite residual loaded directly as the right-hand side of the coarse correction -- and (b) with one
    plane per workgroup (VDN_BATCH_PPW=1), the interpolation over whole boxes (VDN_NDM_IFACE_FACES=0) and the negated copy + zero-filled
    correction of round 2 (VDN_NDM_NEG=1), and (c) with the defaults and the kept descriptor tables bounded to two entries (VDN_KEPT_BOUND=2).  The switches are
    read once per process, hence the child processes."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        from varden_amd import driver
        from varden_amd.capi import default_params
        walls = [[15, 15]] * 3
        prm = default_params(cflfac=0.9)
        levels = driver.VardenAMR.tagged_grids(32, walls, prm, max_levs=3, max_grid_size=32)
        G = driver.VardenAMR(32, levels[0], walls, params=prm, finer_levels=levels[1:], init_iter=1, do_initial_projection=1)
        for _ in range(2):
            G.step()
        h = hashlib.sha256()
        for n in range(3):
            for m in (G.uold[n], G.sold[n], G.p[n], G.gp[n]):
                for f in range(m.nfabs()):
                    h.update(np.ascontiguousarray(m.to_numpy(f)).tobytes())
        print("HASH", h.hexdigest(), G.dt)
    """ % root)
    switches = ("VDN_BATCH_PPW", "VDN_NDM_IFACE_FACES", "VDN_NDM_NEG", "VDN_BATCH_YZ", "VDN_NDF_PAIR", "VDN_BATCH_CHUNK", "VDN_MLCC_FUSE1", "VDN_MLCC_GLUE", "VDN_MLCC_RHO", "VDN_FB_FACES", "VDN_NDF_SEGW", "VDN_GOD_SEGW", "VDN_KEEP_SETS", "VDN_BATCH_FLAT", "VDN_NDM_PROLONG8", "VDN_KEPT_BOUND")
    out = []
    # third run: the kept descriptor tables bounded to two entries each, so that every bound is hit in the MIDDLE of the composite solves of a three-level
    # step (an eviction there must not free the sets the running solve is bound to)
    for extra in ({}, {"VDN_KEPT_BOUND": "2"}, {"VDN_BATCH_PPW": "1", "VDN_NDM_IFACE_FACES": "0", "VDN_NDM_NEG": "1", "VDN_BATCH_YZ": "0", "VDN_NDF_PAIR": "0", "VDN_BATCH_CHUNK": "0", "VDN_MLCC_FUSE1": "0", "VDN_MLCC_GLUE": "0", "VDN_MLCC_RHO": "0", "VDN_FB_FACES": "0", "VDN_NDF_SEGW": "0", "VDN_GOD_SEGW": "0", "VDN_KEEP_SETS": "0", "VDN_BATCH_FLAT": "0", "VDN_NDM_PROLONG8": "0"}):
        env = dict(os.environ)
        for k in switches:
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
    assert out[0] == out[1] == out[2], out


def test_periodic_three_level_hierarchy_is_translation_invariant(gpu):
    """three levels on a domain periodic in x, shifted by half a period so that the boxes of BOTH refined levels sit across the periodic
    boundary: the corrections prolonged linearly into level 2 then read their coarse neighbours through periodic images of another box
    (amr.hip ml_cc_solve, the way up: edge fill + same-level exchange of the source).  Two steps; 1e-8 relative, iteration counts within one."""
    from varden_amd import advance as adv
    from varden_amd import driver
    from varden_amd.capi import default_params
    nc = 16
    phys = [[-1, -1], [15, 15], [15, 15]]

    def bubble(xc):
        def fn(lev, blo, nb, dx):
            g = 3
            ax = [(np.arange(blo[d] - g, blo[d] + nb[d] + g) + 0.5) * dx[d] for d in range(3)]
            X, Y, Z = np.meshgrid(*ax, indexing="ij")
            dxp = (X - xc + 0.5) % 1.0 - 0.5
            r = np.sqrt(dxp ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2)
            s = np.zeros(X.shape + (2,), order="F")
            s[..., 0] = 1.0 + 0.5 * (10.0 - 1.0) * (1.0 - np.tanh(30.0 * (r - 0.1)))
            s[..., 1] = s[..., 0]
            return np.zeros(X.shape + (3,), order="F"), s
        return fn

    prm = lambda: default_params(cflfac=0.9)   # noqa: E731
    A = driver.VardenAMR(nc, [((8, 8, 8), (23, 23, 23))], phys, params=prm(), finer_levels=[[((24, 24, 24), (39, 39, 39))]], init_fn=bubble(0.5), init_iter=1,
                         do_initial_projection=1)
    for _ in range(2):
        A.step()
    ua0 = A.unew[0].to_numpy(0)[3:-3, 3:-3, 3:-3]
    sa2 = A.snew[2].to_numpy(0)[3:-3, 3:-3, 3:-3]
    ua2 = A.unew[2].to_numpy(0)[3:-3, 3:-3, 3:-3]
    dta, ita = A.dt, (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
    A.close()
    B = driver.VardenAMR(nc, [((24, 8, 8), (31, 23, 23)), ((0, 8, 8), (7, 23, 23))], phys, params=prm(),
                         finer_levels=[[((56, 24, 24), (63, 39, 39)), ((0, 24, 24), (7, 39, 39))]], init_fn=bubble(0.0), init_iter=1, do_initial_projection=1)
    for _ in range(2):
        B.step()
    assert abs(B.dt - dta) <= 1e-9 * dta
    assert abs(adv.last_solver_stats("mac")[0] - ita[0]) <= 1 and abs(adv.last_solver_stats("hg")[0] - ita[1]) <= 1, (ita, adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
    ub0 = B.unew[0].to_numpy(0)[3:-3, 3:-3, 3:-3]
    assert np.abs(np.roll(ua0, nc // 2, axis=0) - ub0).max() <= 1e-8 * np.abs(ua0).max()
    for a, mf, nm in ((ua2, B.unew[2], "u"), (sa2, B.snew[2], "s")):
        lo_half, hi_half = mf.to_numpy(0)[3:-3, 3:-3, 3:-3], mf.to_numpy(1)[3:-3, 3:-3, 3:-3]
        assert np.abs(a[:8] - lo_half).max() <= 1e-8 * np.abs(a).max() and np.abs(a[8:] - hi_half).max() <= 1e-8 * np.abs(a).max(), nm
    assert np.abs(ua2[..., 2]).max() > 0
    B.close()


def test_level0_cycles_of_composite_solves_by_colour_agree_bit_for_bit(gpu):
    """round 6: the V-cycle a composite solve runs on level 0 in every FAC iteration takes the level by colour -- the MAC projection's (the density form) and, with viscosity, the three
    velocity solves' (constant face coefficients) -- from 2^23 cells (the 256^3 base of configs[3] / [4]).  Here a 128^3 base with one refined box, visc_coef = 0.001, start-up + one
    step, VDN_MAC_SPLIT_MIN=0 (slabs of 16 planes) against VDN_MAC_SPLIT=0: the same state, bit for bit, and the by-colour form did run."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        from varden_amd import driver, capi
        from varden_amd.capi import default_params
        G = driver.VardenAMR(128, [((48, 48, 40), (111, 111, 103))], [[15, 15]] * 3, params=default_params(cflfac=0.9, visc_coef=0.001), init_shrink=0.1, init_iter=1, do_initial_projection=1)
        G.step()
        h = hashlib.sha256()
        for n in range(2):
            for m in (G.uold[n], G.sold[n], G.p[n], G.gp[n]):
                for f in range(m.nfabs()):
                    h.update(np.ascontiguousarray(m.to_numpy(f)).tobytes())
        print("HASH", h.hexdigest(), G.dt, "FORM", capi.load().vdn_last_mac_level_form())
        G.close()
    """ % root)
    out = []
    # third run: the refined levels' sweeps and residuals READ the face coefficients (VDN_MLCC_RHO=0) instead of forming them from the density (MAC) or taking
    # the constant mu (viscous solves: use_rho == 2 of amr.hip) -- the same bits again
    for extra in ({"VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_SLAB": "16"}, {"VDN_MAC_SPLIT": "0"}, {"VDN_MAC_SPLIT": "0", "VDN_MLCC_RHO": "0"}):
        env = dict(os.environ)
        for k in ("VDN_MAC_SPLIT", "VDN_MAC_SPLIT_MIN", "VDN_MAC_SLAB", "VDN_MLCC_RHO"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0].split())
    assert out[0][1] == out[1][1] == out[2][1] and out[0][2] == out[1][2] == out[2][2], out
    assert out[0][4] == "1" and out[1][4] == "0", out          # (the last cell-centred solve of the step: a viscous composite one; its level-0 cycles by colour / interleaved)
