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


# ---- 2-D hierarchies: the 2-D problem run as its z-uniform copy on the 3-D machinery (round 6; driver.VardenAMR: extrude2d, include/varden_amd.h: vdn_set_extruded_2d) -----------
def _prm_pair(bc, **kw):
    from varden_amd.capi import default_params
    out = []
    for dm in (2, 3):
        p = default_params(dm=dm, cflfac=0.9, **kw) if dm == 2 else default_params(cflfac=0.9, **kw)
        for d in range(2):
            for s in range(2):
                if bc[d][s] == 11:
                    [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                    p.rho_bc[d][s] = 1.0
                    p.trac_bc[d][s] = 0.5
        out.append(p)
    return out


@pytest.mark.parametrize("name,bc,prob,visc", [("bubble-walls", [[15, 15], [15, 15]], 1, 0.0), ("bubble-periodic-x-viscous", [[-1, -1], [15, 15]], 1, 0.001),
                                                ("blob-inflow-outflow-walls", [[11, 12], [15, 15]], 2, 0.001), ("blob-inflow-outflow-slip", [[11, 12], [14, 14]], 2, 0.0),
                                                ("outflow-both-x", [[12, 12], [15, 15]], 2, 0.0), ("outflow-y", [[15, 15], [12, 12]], 2, 0.0),
                                                ("rayleigh-taylor-periodic-x", [[-1, -1], [15, 15]], 3, 0.01)])
def test_extruded_copy_reproduces_the_2d_path(gpu, name, bc, prob, visc):
    """The basis of the 2-D hierarchies: a 2-D run (dim2.hip: velpred_2d / mkflux_2d, 5- and 9-point solvers, checked against oracle/vo_2d.c above) against the SAME data
    extruded along a periodic z -- n x n x nz cells, w = 0, gravity along y -- through the 3-D kernels: with nothing varying along z every 3-D operator is its 2-D
    counterpart (the red-black order and the solvers' tolerances leave 1e-13).  The one place where the reference's two restatements differ -- velpred_3d clamps the normal
    velocity of a hi-x OUTLET with min(), velpred.f90:2075, velpred_2d with max(), :305 -- takes the 2-D rule (vdn_set_extruded_2d); without it the outflow corners differ by
    5e-4 after four steps (tools/probes/extruded2d_inout_probe.py).  Every boundary pair of the four 2-D inputs of exec/test, four steps after the start-up sequence: dt bit for
    bit or to 1e-12, u, rho, tracer and grad p of plane k = 0 to 1e-10 of their scale, |w| and the spread over z below 1e-10."""
    from varden_amd import driver
    n, nz, nsteps = 32, 8, 4
    p2, p3 = _prm_pair(bc, visc_coef=visc)
    kw = dict(prob_type=prob, init_shrink=0.1, init_iter=1)
    G2 = driver.Varden(n, [bc[0], bc[1], [0, 0]], p2, **kw)
    for _ in range(nsteps):
        G2.step()
    u2, s2, g2, dt2 = G2.gather_valid(G2.uold[0])[:, :, 0, :], G2.gather_valid(G2.sold[0])[:, :, 0, :], G2.gather_valid(G2.gp[0])[:, :, 0, :], G2.dt
    G2.close()
    u0_2, s0_2 = driver.initdata_numpy((n, n), [1.0 / n] * 2, prob, 3, 2, dm=2)
    u0 = np.zeros((n + 6, n + 6, nz + 6, 3), order="F")
    s0 = np.zeros((n + 6, n + 6, nz + 6, 2), order="F")
    u0[..., :2] = u0_2[:, :, 0, None, :]
    s0[...] = s0_2[:, :, 0, None, :]
    G3 = driver.Varden((n, n, nz), [bc[0], bc[1], [-1, -1]], p3, prob_hi=(1.0, 1.0, nz / float(n)), u0=u0, s0=s0, grav_dir=1, extruded2d=True, **kw)
    for _ in range(nsteps):
        G3.step()
    u3, s3, g3 = G3.gather_valid(G3.uold[0]), G3.gather_valid(G3.sold[0]), G3.gather_valid(G3.gp[0])
    assert abs(G3.dt - dt2) <= 1e-12 * dt2
    G3.close()
    su, sg = max(np.abs(u2).max(), 1e-3), max(np.abs(g2).max(), 1e-3)
    assert np.abs(u3[:, :, 0, :2] - u2).max() <= 1e-10 * su, name
    assert np.abs(s3[:, :, 0, :] - s2).max() <= 1e-10 * np.abs(s2).max(), name
    assert np.abs(g3[:, :, 0, :2] - g2).max() <= 1e-8 * sg, name
    assert np.abs(u3[..., 2]).max() <= 1e-10 * su and np.abs(u3 - u3[:, :, :1, :]).max() <= 1e-10 * su and np.abs(s3 - s3[:, :, :1, :]).max() <= 1e-10


@pytest.mark.parametrize("name,bc,prob,max_levs", [("bubble-walls", [[15, 15], [15, 15]], 1, 3), ("bubble-periodic-x", [[-1, -1], [15, 15]], 1, 2)])
def test_extruded_hierarchy_against_the_box_list_oracle(gpu, oracle, name, bc, prob, max_levs):
    """A tagged 2-D hierarchy as its z-uniform copy, held against the CPU oracle's box-list hierarchies (oracle/vo_amr.c) on the same boxes: 32 x 32 x 8 cells of level 0, the
    levels tagged from initdata_2d (tag_boxes.f90:65-84, the same thresholds in tag_boxes_2d), viscous, start-up + two steps.  The oracle keeps no periodic images of refined boxes, so its copy
    stands between SLIP WALLS along z -- the same z-uniform solution (w = 0 on the wall, nothing to reflect) -- and the library runs both forms: against the oracle with the walls
    (dt bit for bit, equal FAC counts, u / rho to 1e-9), and with the periodic z of the product path against its own wall run (plane k = 0 to 1e-10: the refined boxes talk
    to their own periodic images along z there)."""
    from varden_amd import advance as adv, driver
    from varden_amd.capi import default_params
    vo = oracle
    n, nz = 32, 8
    prm = lambda: default_params(cflfac=0.9, visc_coef=0.001)   # noqa: E731
    levels = driver.VardenAMR.tagged_grids((n, n), bc, prm(), prob_type=prob, max_levs=max_levs, max_grid_size=32, extrude2d=nz)
    assert len(levels) == max_levs - 1
    for lb in levels:                                       # the boxes are columns: z-uniform tags cluster into z-uniform footprints
        foot = {}
        for lo, hi in lb:
            foot.setdefault((lo[0], lo[1], hi[0], hi[1]), []).append((lo[2], hi[2]))
        top = max(hi[2] for _, hi in lb)
        for f, zs in foot.items():
            zs.sort()
            assert zs[0][0] == 0 and zs[-1][1] == top and all(zs[i][1] + 1 == zs[i + 1][0] for i in range(len(zs) - 1)), (f, zs)
    kw = dict(prob_type=prob, init_shrink=0.1, init_iter=1, do_initial_projection=1)
    init = driver.extruded_initdata(prob, 2)
    phys3w = [bc[0], bc[1], [14, 14]]
    O = vo.SimML((n, n, nz), levels, phys3w, prm=prm(), init_fn=init, grav_dir=1, **kw)
    Gw = driver.VardenAMR((n, n), levels[0], bc, params=prm(), finer_levels=levels[1:], extrude2d=nz, extrude_zbc=[14, 14], **kw)
    assert Gw.dt == O.dt
    for _ in range(2):
        O.step(); Gw.step()
        assert Gw.dt == O.dt
        assert adv.last_solver_stats("mac")[0] == O.mgstat[0].cycles and adv.last_solver_stats("hg")[0] == O.mgstat[1].cycles
    planes_w = [Gw.slice2d(Gw.uold), Gw.slice2d(Gw.sold)]
    for lev in range(Gw.nlev):
        olo = O.levels[lev].lo
        for li, gi in enumerate(Gw.local[lev]):
            lo, hi = Gw.boxes[lev][gi]
            sl = tuple(slice(lo[d] - olo[d], hi[d] - olo[d] + 1) for d in range(3))
            a = Gw.uold[lev].to_numpy(li)[3:-3, 3:-3, 3:-3]
            b = O.uold[lev].valid()[sl]
            assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-3), (name, lev, gi)
            a = Gw.sold[lev].to_numpy(li)[3:-3, 3:-3, 3:-3]
            b = O.sold[lev].valid()[sl]
            assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), (name, lev, gi)
    Gw.close()
    Gp = driver.VardenAMR((n, n), levels[0], bc, params=prm(), finer_levels=levels[1:], extrude2d=nz, **kw)
    for _ in range(2):
        Gp.step()
    for wl, pl in zip(planes_w, [Gp.slice2d(Gp.uold), Gp.slice2d(Gp.sold)]):
        for lev in range(Gp.nlev):
            m = np.isfinite(wl[lev])
            assert (m == np.isfinite(pl[lev])).all()
            assert np.abs(wl[lev][m] - pl[lev][m]).max() <= 1e-10 * max(np.abs(wl[lev][m]).max(), 1e-3), (name, lev)
    Gp.close()


@pytest.mark.parametrize("inputs_name,steps", [("inputs_bubble_2d", 6), ("inputs_2d-regt", 6), ("inputs_advect_2d", 6), ("inputs_RayleighTaylor_2d", 36)])
def test_2d_inputs_of_the_reference_run_as_extruded_hierarchies(gpu, tmp_path, inputs_name, steps):
    """The four 2-D inputs of exec/test (all adaptive: max_levs 3-4, regrid_int 1-2, viscous) through varden_amd.inputs: tagging and clustering of the initial data, start-up,
    time loop with regridding, plot files of the 3-D copy.  Every step: both composite solves converge, |w| and the spread over z stay at round-off, the bubble stays
    mirror-symmetric in x up to the truncation error of its (unsymmetric) box layout; the grids were rebuilt; the finest level sits where the density varies."""
    import os
    from varden_amd import advance as adv, inputs
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs", inputs_name)).read()
    seen = []

    def rep(G):
        assert adv.last_solver_stats("mac")[2] <= 1e-10 * adv.last_solver_stats("mac")[1] and adv.last_solver_stats("hg")[0] < 60
        seen.append((G.istep, G.nregrids, [len(b) for b in G.boxes]))
    nl, G = inputs.run(text, nsteps=steps, report=rep, outdir=str(tmp_path))
    assert G.extrude2d and G.nlev == int(nl["max_levs"]) and G.nregrids >= steps // max(int(nl["regrid_int"]), 1) - 1
    umax = max(np.abs(G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3, :2]).max() for n in range(G.nlev) for i in range(G.uold[n].nfabs()))
    assert np.isfinite(umax) and umax > 0
    for n in range(G.nlev):
        for i in range(G.uold[n].nfabs()):
            u, s = G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3], G.sold[n].to_numpy(i)[3:-3, 3:-3, 3:-3]
            assert np.abs(u[..., 2]).max() <= 1e-10 * umax and np.abs(u - u[:, :, :1]).max() <= 1e-10 * umax and np.abs(s - s[:, :, :1]).max() <= 1e-10
    # proper nesting, periodic faces included: the cells of level n - 1 under a box of level n, grown by two, belong to level n - 1 -- across a periodic x face their
    # periodic images (make_new_grids' nesting map wraps there; before round 6 it did not, and inputs_RayleighTaylor_2d -- interface along a periodic x, regrid_int = 1 --
    # met an improperly nested level at its 31st regrid and failed with a non-finite right-hand side)
    perx = int(nl["bcx_lo"]) == -1
    for n in range(2, G.nlev):
        par = np.zeros((G.ncs[0] << (n - 1), G.ncs[1] << (n - 1)), dtype=bool)
        for lo, hi in G.boxes[n - 1]:
            par[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1] = True
        need = np.zeros_like(par)
        for lo, hi in G.boxes[n]:
            need[lo[0] // 2:hi[0] // 2 + 1, lo[1] // 2:hi[1] // 2 + 1] = True
        grown = need.copy()
        for ax in (0, 1):
            for sft in (1, 2, -1, -2):
                r = np.roll(need, sft, axis=ax)
                if not (ax == 0 and perx):                      # a wall: nothing comes around
                    idx = [slice(None)] * 2
                    idx[ax] = slice(0, sft) if sft > 0 else slice(sft, None)
                    r[tuple(idx)] = False
                grown |= r
        assert (par | ~grown).all(), "level %d of %s is not nested in level %d" % (n, inputs_name, n - 1)
    rho = G.slice2d(G.sold)
    if inputs_name in ("inputs_bubble_2d", "inputs_2d-regt"):
        r0 = rho[0][..., 0]
        assert np.abs(r0 - r0[::-1, :]).max() <= 5e-3              # (the clustered boxes are not mirror images of each other: symmetric up to the coarse-fine truncation error)
    fin = np.isfinite(rho[-1][..., 0])
    assert fin.any() and (np.abs(rho[0][..., 0] - (1.0 if inputs_name != "inputs_RayleighTaylor_2d" else rho[0][0, 0, 0])) > 0.05).any()
    assert any(f.startswith(str(tmp_path)) for f in G.files_written)
    G.close()


def test_extruded_hierarchy_follows_the_fine_2d_run(gpu):
    """No 2-D oracle for hierarchies exists; the physical check beside the parity chain above: the bubble on a 32^2 base with its tagged region refined (two levels, viscous, walls), run as
    the extruded copy, against the ONE-level 2-D runs (dim2.hip) at 32^2 and at 64^2 with the same fixed dt, twenty steps.  Where the hierarchy is refined it must follow the fine
    run: u to 1e-3 (0.3 % of max|u|; the coarse run is 2.5e-2 away on the same cells), rho to 1e-4 (coarse run: 0.16) -- measured 3.3e-4 and 1.1e-5 (tools/probes/extruded2d_vs_fine_probe.py)."""
    from varden_amd import driver
    from varden_amd.capi import default_params
    bc = [[15, 15], [15, 15]]
    dt, nsteps = 2.0e-3, 20
    runs = {}
    for n in (32, 64):
        G = driver.Varden(n, [bc[0], bc[1], [0, 0]], default_params(dm=2, cflfac=0.9, visc_coef=0.001), prob_type=1, init_shrink=1.0, init_iter=1, fixed_dt=dt)
        for _ in range(nsteps):
            G.step()
        runs[n] = (G.gather_valid(G.uold[0])[:, :, 0, :], G.gather_valid(G.sold[0])[:, :, 0, :])
        G.close()
    levels = driver.VardenAMR.tagged_grids((32, 32), bc, default_params(cflfac=0.9, visc_coef=0.001), prob_type=1, max_levs=2, max_grid_size=32, extrude2d=8)
    G = driver.VardenAMR((32, 32), levels[0], bc, params=default_params(cflfac=0.9, visc_coef=0.001), prob_type=1, init_shrink=1.0, init_iter=1, do_initial_projection=1,
                         extrude2d=8, fixed_dt=dt)
    for _ in range(nsteps):
        G.step()
    u, s = G.slice2d(G.uold), G.slice2d(G.sold)
    G.close()
    m = np.isfinite(u[1][..., 0])
    assert 0.1 * m.size < m.sum() < 0.6 * m.size
    (uf, sf), (uc, sc) = runs[64], runs[32]
    up = lambda a: np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)      # noqa: E731
    du, dr = np.abs(u[1][..., :2][m] - uf[m]).max(), np.abs(s[1][..., 0][m] - sf[..., 0][m]).max()
    cu, cr = np.abs(up(uc)[m] - uf[m]).max(), np.abs(up(sc)[..., 0][m] - sf[..., 0][m]).max()
    assert du <= 1e-3 and dr <= 1e-4 and du <= 0.05 * cu and dr <= 0.01 * cr, (du, dr, cu, cr)
