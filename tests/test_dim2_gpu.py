"""dm = 2 (BASELINE.json configs[0]: "2D 128^2 single-level bubble", the reference's own CPU-runnable case): parity of
the HIP path with the CPU oracle's restatement of velpred_2d / mkflux_2d / update_2d / the 2-D projection kernels.
Bit-exact for the Godunov kernels and the update; 1e-9 relative for whole steps (the two multigrids run the same
algorithm in the same order, in practice ~1e-15)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import assert_bits

pytestmark = pytest.mark.gpu

BC2 = {
    "walls": [[15, 15], [15, 15]],
    "slip": [[14, 14], [14, 14]],
    "periodic": [[-1, -1], [-1, -1]],
    "inout": [[11, 12], [14, 15]],          # inlet x-lo, outlet x-hi (exec/test/inputs_advect_2d style)
    "outin-y": [[15, 14], [12, 11]],        # outlet y-lo, inlet y-hi
}


def params2(phys, **kw):
    from varden_amd.capi import default_params
    p = default_params(dm=2, **kw)
    for d in range(2):
        for s in range(2):
            if phys[d][s] == 11:
                [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                p.rho_bc[d][s] = 1.0
                p.trac_bc[d][s] = 0.5
    return p


class Case2:
    def __init__(self, n, phys, seed=0, **kw):
        from oracle import voracle as vo
        from varden_amd import boxlib as bl
        self.vo, self.bl = vo, bl
        self.n = (n[0], n[1], 1)
        self.phys3 = [list(phys[0]), list(phys[1]), [0, 0]]
        self.prm = params2(phys, **kw)
        bl.initialize(self.prm, 0, 1, 0)
        self.rng = np.random.default_rng(seed)
        self.lo, self.hi = (0, 0, 0), (n[0] - 1, n[1] - 1, 0)
        self.pmask = [1 if phys[d][0] == -1 else 0 for d in range(2)] + [0]
        self.obc = vo.make_bc(self.phys3, 2, self.prm.nscal)
        self.opm = vo.ivec(self.pmask)
        self.mla = bl.MLLayout([(self.lo, self.hi)], [[(self.lo, self.hi)]], pmask=self.pmask)
        self.bct = bl.BCTower(self.mla, self.phys3)
        self.dx = [1.0 / max(n)] * 2
        self.odx = vo.dvec(self.dx + [1.0])
        self._mfs = []

    def ofab(self, ng, nc, nodal=(0, 0, 0), val=0.0):
        return self.vo.Fab(self.lo, self.hi, ng, nc, nodal, val, dm=2)

    def gmf(self, ofab):
        mf = self.bl.MultiFab(self.mla, 0, ofab.nc, ofab.ng, ofab.nodal)
        mf.from_numpy(ofab.a)
        self._mfs.append(mf)
        return mf

    def smooth(self, f, amp, base=0.0):
        nx, ny = f.a.shape[:2]
        X, Y = np.meshgrid((np.arange(nx) + 0.5) / nx, (np.arange(ny) + 0.5) / ny, indexing="ij")
        for c in range(f.nc):
            v = np.zeros((nx, ny))
            for _ in range(4):
                k = self.rng.integers(1, 4, size=2); ph = self.rng.uniform(0, 2 * np.pi, size=2)
                v += self.rng.uniform(-1, 1) * np.sin(2 * np.pi * k[0] * X + ph[0]) * np.sin(2 * np.pi * k[1] * Y + ph[1])
            v += 0.05 * self.rng.standard_normal((nx, ny))
            f.a[:, :, 0, c] = base + amp * v

    def random_state(self):
        L = self.vo.lib()
        u, s = self.ofab(3, 2), self.ofab(3, self.prm.nscal)
        self.smooth(u, 1.0); self.smooth(s, 0.3, 2.0)
        for f in (u, s):
            L.vo_fill_boundary(f.ref, self.opm)
        L.vo_physbc(u.ref, 0, 0, 2, C.byref(self.obc), C.byref(self.prm))
        L.vo_physbc(s.ref, 0, 2, self.prm.nscal, C.byref(self.obc), C.byref(self.prm))
        return u, s

    def close(self):
        for m in self._mfs:
            m.destroy()
        self.bct.destroy(); self.mla.destroy()


@pytest.mark.parametrize("bcname", list(BC2))
@pytest.mark.parametrize("minion", [0, 1])
def test_godunov_2d_bits(gpu, oracle, bcname, minion):
    """velpred_2d, mkflux_2d (velocity: convective; scalars: comp 1 conservative) and update_2d, bit for bit"""
    from varden_amd import advance as adv
    vo = oracle
    K = Case2((24, 20), BC2[bcname], seed=3, use_minion=minion)
    L = vo.lib()
    u, s = K.random_state()
    f2, fs, rhs = K.ofab(1, 2), K.ofab(1, 2), K.ofab(1, 1)
    K.smooth(f2, 0.5); K.smooth(fs, 0.2); K.smooth(rhs, 0.1)
    for f in (f2, fs, rhs):
        L.vo_fill_boundary(f.ref, K.opm)
    nd = [(1, 0, 0), (0, 1, 0)]
    um = [K.ofab(1, 1, nd[d], 1e20) for d in range(2)]
    dt = 0.3 * K.dx[0]
    gu, gs, gf2, gfs, grhs = K.gmf(u), K.gmf(s), K.gmf(f2), K.gmf(fs), K.gmf(rhs)
    gum = [K.gmf(m) for m in um]
    L.vo2_velpred(u.ref, vo.fab_ptr_array(um), f2.ref, K.odx, C.c_double(dt), C.byref(K.obc), C.byref(K.prm))
    for m in um:
        L.vo_fill_boundary(m.ref, K.opm)
    adv.velpred(gu, gum, gf2, K.dx, dt, K.bct)
    for d in range(2):
        a, b = gum[d].to_numpy(), um[d].a
        assert_bits(a[1:-1, 1:-1], b[1:-1, 1:-1], "umac[%d] %s" % (d, bcname))
    for is_vel, st, fo, nc, cons in ((True, u, f2, 2, [0, 0]), (False, s, fs, 2, [1, 0])):
        se = [K.ofab(0, nc, nd[d]) for d in range(2)]; fl = [K.ofab(0, nc, nd[d]) for d in range(2)]
        gse = [K.gmf(m) for m in se]; gfl = [K.gmf(m) for m in fl]
        gst, gfo = (gu, gf2) if is_vel else (gs, gfs)
        L.vo2_mkflux(st.ref, vo.fab_ptr_array(se), vo.fab_ptr_array(fl), vo.fab_ptr_array(um), fo.ref, rhs.ref, K.odx, C.c_double(dt),
                     1 if is_vel else 0, vo.ivec(cons), 0 if is_vel else 2, C.byref(K.obc), C.byref(K.prm))
        adv.mkflux(gst, gse, gfl, gum, gfo, grhs, K.dx, dt, K.bct, is_vel, cons)
        for d in range(2):
            assert_bits(gse[d].to_numpy(), se[d].a, "sedge[%d] vel=%s %s" % (d, is_vel, bcname))
            if not is_vel:
                assert_bits(gfl[d].to_numpy()[..., 0], fl[d].a[..., 0], "flux[%d] %s" % (d, bcname))
        sn = st.like(); gsn = K.gmf(sn)
        L.vo2_update(st.ref, vo.fab_ptr_array(um), vo.fab_ptr_array(se), vo.fab_ptr_array(fl), fo.ref, sn.ref, K.odx, C.c_double(dt),
                     1 if is_vel else 0, vo.ivec(cons))
        adv.update(gst, gum, gse, gfl, gfo, gsn, K.dx, dt, is_vel, cons, K.bct)
        assert_bits(gsn.to_numpy()[3:-3, 3:-3], sn.a[3:-3, 3:-3], "update vel=%s %s" % (is_vel, bcname))
    K.close()


def run_pair2(n, phys, nsteps, prob_type=1, **kw):
    from oracle import voracle as vo
    from varden_amd import driver
    phys3 = [list(phys[0]), list(phys[1]), [0, 0]]
    O = vo.Sim(n, phys3, params2(phys, cflfac=0.9, **kw), prob_type=prob_type, init_shrink=0.1, init_iter=2, dm=2)
    G = driver.Varden(n, phys3, params2(phys, cflfac=0.9, **kw), prob_type=prob_type, init_shrink=0.1, init_iter=2)
    assert G.dt == O.dt
    for _ in range(nsteps):
        O.step(); G.step()
        assert G.dt == O.dt, "dt diverged: %r vs %r" % (G.dt, O.dt)
    return O, G


def check_pair2(O, G, name):
    for nm, gm, om in (("u", G.unew[0], O.unew), ("s", G.snew[0], O.snew)):
        a, b = G.gather_valid(gm), om.valid()
        scale = max(np.abs(b).max(), 1e-300)
        assert np.abs(a - b).max() <= 1e-9 * scale, "%s: %s differs by %.3e (scale %.3e)" % (name, nm, np.abs(a - b).max(), scale)
    a, b = G.p[0].to_numpy()[1:-1, 1:-1], O.p.valid()
    assert np.abs((a - a.mean()) - (b - b.mean())).max() <= 1e-6 * max(np.abs(b - b.mean()).max(), 1e-300)


@pytest.mark.parametrize("name,bc,prob", [("bubble-walls", "walls", 1), ("bubble-periodic", "periodic", 1), ("blob-inout", "inout", 2)])
def test_advance_2d_parity(gpu, name, bc, prob):
    O, G = run_pair2(32, BC2[bc], 4, prob_type=prob)
    check_pair2(O, G, name)
    G.close()


@pytest.mark.parametrize("dtype", [1, 2])
def test_advance_2d_viscous(gpu, dtype):
    """exec/test/inputs_bubble_2d has visc_coef > 0: explicit diffusive term + implicit solves, Crank-Nicolson and backward Euler"""
    O, G = run_pair2(32, BC2["walls"], 3, visc_coef=0.01, diff_coef=0.005, diffusion_type=dtype)
    check_pair2(O, G, "viscous-%d" % dtype)
    G.close()


def test_config0_bubble_128(gpu):
    """BASELINE.json configs[0]: 2-D 128^2 bubble (exec/test/inputs_bubble_2d with n_cell = 128, max_levs = 1, all walls,
    cflfac 0.9, init_shrink 0.1, init_iter 1, inviscid): 5 steps against the oracle, and the size-independent properties"""
    from varden_amd import advance as adv
    O, G = run_pair2(128, BC2["walls"], 5)
    check_pair2(O, G, "config0")
    s = G.gather_valid(G.snew[0])[:, :, 0, 0]
    u = G.gather_valid(G.unew[0])[:, :, 0, :]
    assert np.abs(s - s[::-1, :]).max() <= 1e-9 * np.abs(s).max()          # mirror symmetry about x = 1/2
    assert np.abs(u[..., 0] + u[::-1, :, 0]).max() <= 1e-9 * np.abs(u).max()
    assert adv.last_solver_stats("mac")[0] < 30 and adv.last_solver_stats("hg")[0] < 40
    G.close()
