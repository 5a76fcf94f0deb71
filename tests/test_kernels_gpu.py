"""GPU parity (through the C-ABI) of every per-box kernel of the hot path against the CPU oracle, on
seeded inputs and for every boundary-condition family the reference handles.  The bar for the
Godunov / streaming kernels is BIT-EXACT: the HIP kernels keep the reference's expression order and are
built with fp-contract off (tolerance 0 ulp)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import BC_SETS, Case, assert_bits

pytestmark = pytest.mark.gpu


def face_fabs(case, ng, nc, val=0.0):
    return [case.ofab(ng, nc, tuple(1 if t == d else 0 for t in range(3)), val) for d in range(3)]


@pytest.mark.parametrize("bcname", list(BC_SETS))
@pytest.mark.parametrize("order", [4, 2, 0])
def test_slope(gpu, oracle, bcname, order):
    from varden_amd import advance as adv
    case = Case((16, 12, 8), BC_SETS[bcname], seed=1, slope_order=order)
    u, s = case.random_state()
    for src, bccomp in ((u, 0), (s, 3)):
        gsrc = case.gmf(src)
        for d in range(3):
            osl = case.ofab(1, src.nc)
            oracle.lib().vo_slope(src.ref, osl.ref, d, src.nc, bccomp, C.byref(case.obc), order)
            gsl = case.gmf(case.ofab(1, src.nc))
            adv.slope(gsrc, gsl, d, bccomp, case.bct)
            assert_bits(gsl.to_numpy(), osl.a, "slope dir %d bc %s order %d" % (d, bcname, order))
    case.close()


def test_slope_minimum_width(gpu, oracle):
    """boxes exactly 4 cells wide: the lo and hi one-sided corrections interleave (slope.f90:243-283)"""
    from varden_amd import advance as adv
    case = Case((4, 4, 4), BC_SETS["walls"], seed=5)
    u, _ = case.random_state()
    gu = case.gmf(u)
    for d in range(3):
        osl = case.ofab(1, 3)
        oracle.lib().vo_slope(u.ref, osl.ref, d, 3, 0, C.byref(case.obc), 4)
        gsl = case.gmf(case.ofab(1, 3))
        adv.slope(gu, gsl, d, 0, case.bct)
        assert_bits(gsl.to_numpy(), osl.a, "slope width-4 dir %d" % d)
    case.close()


# (16, 8, 32): every dx = 1 / n is a power of two -- the fused marches then scale by 1 / dx instead of dividing (godunov.hip, PW2 kernels: the same
# doubles); (16, 12, 8) / (12, 16, 8) keep the dividing kernels under test
@pytest.mark.parametrize("bcname", list(BC_SETS))
@pytest.mark.parametrize("minion", [0, 1])
@pytest.mark.parametrize("n", [(16, 12, 8), (16, 8, 32)])
def test_velpred(gpu, oracle, bcname, minion, n):
    from varden_amd import advance as adv
    case = Case(n, BC_SETS[bcname], seed=2, use_minion=minion)
    u, _ = case.random_state()
    force = case.ofab(1, 3)
    force.a[...] = case.rng.standard_normal(force.a.shape)
    oracle.lib().vo_fill_boundary(force.ref, case.opm)      # periodic images, as ml_restrict_and_fill leaves them
    dt = 0.4 * min(case.dx)
    oum = face_fabs(case, 1, 1, 1.0e20)
    oracle.lib().vo_velpred(u.ref, oracle.fab_ptr_array(oum), force.ref, case.odx, C.c_double(dt), C.byref(case.obc), C.byref(case.prm))
    for f in oum:
        oracle.lib().vo_fill_boundary(f.ref, case.opm)
    gum = [case.gmf(f) for f in face_fabs(case, 1, 1, 1.0e20)]
    adv.velpred(case.gmf(u), gum, case.gmf(force), case.dx, dt, case.bct)
    for d in range(3):
        assert_bits(gum[d].to_numpy(), oum[d].a, "umac[%d] bc %s minion %d" % (d, bcname, minion))
    case.close()


@pytest.mark.parametrize("bcname", list(BC_SETS))
@pytest.mark.parametrize("is_vel", [0, 1])
@pytest.mark.parametrize("minion", [0, 1])
@pytest.mark.parametrize("n", [(12, 16, 8), (8, 16, 32)])
def test_mkflux(gpu, oracle, bcname, is_vel, minion, n):
    from varden_amd import advance as adv
    case = Case(n, BC_SETS[bcname], seed=3, use_minion=minion)
    u, s = case.random_state()
    src = u if is_vel else s
    nc = src.nc
    is_cons = [0, 0, 0] if is_vel else [1, 0]
    force = case.ofab(1, nc)
    force.a[...] = case.rng.standard_normal(force.a.shape)
    mac_rhs = case.ofab(1, 1)
    mac_rhs.a[...] = 0.1 * case.rng.standard_normal(mac_rhs.a.shape)
    oracle.lib().vo_fill_boundary(force.ref, case.opm)
    oracle.lib().vo_fill_boundary(mac_rhs.ref, case.opm)
    # a MAC velocity field with filled (periodic) ghost faces and 1e20 elsewhere, like the reference's
    oum = face_fabs(case, 1, 1, 1.0e20)
    for d, f in enumerate(oum):
        v = f.valid()
        v[...] = case.rng.uniform(-1, 1, size=v.shape)
        if case.pmask[d]:
            sl = [slice(None)] * 4
            sl_lo, sl_hi = list(sl), list(sl)
            sl_lo[d], sl_hi[d] = 0, -1
            v[tuple(sl_hi)] = v[tuple(sl_lo)]          # periodic: the hi face is the lo face
        oracle.lib().vo_fill_boundary(f.ref, case.opm)
    dt = 0.4 * min(case.dx)
    osedge, oflux = face_fabs(case, 0, nc), face_fabs(case, 0, nc)
    oracle.lib().vo_mkflux(src.ref, oracle.fab_ptr_array(osedge), oracle.fab_ptr_array(oflux), oracle.fab_ptr_array(oum), force.ref,
                           mac_rhs.ref, case.odx, C.c_double(dt), is_vel, oracle.ivec(is_cons), 0 if is_vel else 3,
                           C.byref(case.obc), C.byref(case.prm))
    gsedge = [case.gmf(f) for f in face_fabs(case, 0, nc)]
    gflux = [case.gmf(f) for f in face_fabs(case, 0, nc)]
    adv.mkflux(case.gmf(src), gsedge, gflux, [case.gmf(f) for f in oum], case.gmf(force), case.gmf(mac_rhs), case.dx, dt, case.bct,
               is_vel, is_cons)
    for d in range(3):
        assert not np.isnan(osedge[d].a).any(), "oracle produced NaN (read of an unset intermediate)"
        assert_bits(gsedge[d].to_numpy(), osedge[d].a, "sedge[%d] bc %s is_vel %d" % (d, bcname, is_vel))
        assert_bits(gflux[d].to_numpy(), oflux[d].a, "flux[%d] bc %s is_vel %d" % (d, bcname, is_vel))
    case.close()


@pytest.mark.parametrize("boussinesq", [0, 1])
@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout"])
def test_forces_update_halftime(gpu, oracle, bcname, boussinesq):
    """boussinesq = 1: the external force is scaled by the second scalar on the VALID cells only (mkforce.f90:160-177), not on the
    one-cell face halo that follows (:186-234)"""
    from varden_amd import advance as adv
    case = Case((8, 12, 16), BC_SETS[bcname], seed=4, boussinesq=boussinesq)
    L = oracle.lib()
    u, s = case.random_state()
    ns = case.prm.nscal
    gp, ext, exts = case.ofab(1, 3), case.ofab(1, 3), case.ofab(1, ns)
    for f in (gp, ext, exts):
        f.a[...] = case.rng.standard_normal(f.a.shape)
    # mkvelforce + ghost fill
    ovf = case.ofab(1, 3)
    L.vo_mkvelforce(ovf.ref, ext.ref, gp.ref, s.ref, None, C.c_double(1.0), C.byref(case.prm))
    L.vo_fill_boundary(ovf.ref, case.opm)
    for c in range(3):
        L.vo_physbc(ovf.ref, c, case.obc.extrap_comp, 1, C.byref(case.obc), C.byref(case.prm))
    gvf = case.gmf(case.ofab(1, 3))
    gs, gu = case.gmf(s), case.gmf(u)
    adv.mkvelforce(gvf, case.gmf(ext), gs, case.gmf(gp), None, 1.0, case.bct)
    assert_bits(gvf.to_numpy(), ovf.a, "vel_force " + bcname)
    # mkscalforce
    osf = case.ofab(1, ns)
    L.vo_mkscalforce(osf.ref, exts.ref, None, C.c_double(1.0), C.byref(case.prm))
    L.vo_fill_boundary(osf.ref, case.opm)
    for c in range(ns):
        L.vo_physbc(osf.ref, c, case.obc.extrap_comp, 1, C.byref(case.obc), C.byref(case.prm))
    gsf = case.gmf(case.ofab(1, ns))
    adv.mkscalforce(gsf, case.gmf(exts), None, 1.0, case.bct)
    assert_bits(gsf.to_numpy(), osf.a, "scal_force " + bcname)
    # update (scalars: conservative + convective; velocity) incl. the ghost fill of the result
    oum = face_fabs(case, 1, 1, 0.0)
    for f in oum:
        f.a[...] = case.rng.uniform(-1, 1, size=f.a.shape)
    gum = [case.gmf(f) for f in oum]
    dt = 0.3 * min(case.dx)
    for is_vel, src, gsrc, frc, gfrc, bccomp in ((0, s, gs, osf, gsf, 3), (1, u, gu, ovf, gvf, 0)):
        nc = src.nc
        ose, ofl = face_fabs(case, 0, nc), face_fabs(case, 0, nc)
        for f in ose + ofl:
            f.a[...] = case.rng.standard_normal(f.a.shape)
        is_cons = [1, 0] if not is_vel else [0, 0, 0]
        onew = case.ofab(3, nc)
        L.vo_update(src.ref, oracle.fab_ptr_array(oum), oracle.fab_ptr_array(ose), oracle.fab_ptr_array(ofl), frc.ref, onew.ref,
                    case.odx, C.c_double(dt), is_vel, oracle.ivec(is_cons))
        L.vo_fill_boundary(onew.ref, case.opm)
        L.vo_physbc(onew.ref, 0, bccomp, nc, C.byref(case.obc), C.byref(case.prm))
        gnew = case.gmf(case.ofab(3, nc))
        adv.update(gsrc, gum, [case.gmf(f) for f in ose], [case.gmf(f) for f in ofl], gfrc, gnew, case.dx, dt, is_vel, is_cons, case.bct)
        assert_bits(gnew.to_numpy(), onew.a, "update is_vel=%d %s" % (is_vel, bcname))
    # make_at_halftime
    s2 = s.copy()
    s2.a[...] += 0.1
    orh = case.ofab(1, 3)
    L.vo_make_at_halftime(orh.ref, 0, s.ref, s2.ref, 0)
    L.vo_fill_boundary(orh.ref, case.opm)
    L.vo_physbc(orh.ref, 0, 3, 1, C.byref(case.obc), C.byref(case.prm))
    grh = case.gmf(case.ofab(1, 3))
    adv.make_at_halftime(grh, gs, case.gmf(s2), 0, 0, case.bct)
    assert_bits(grh.to_numpy()[..., 0], orh.a[..., 0], "rhohalf " + bcname)
    case.close()


@pytest.mark.parametrize("bcname", list(BC_SETS))
def test_physbc_fill_boundary_estdt(gpu, oracle, bcname):
    from varden_amd import advance as adv
    case = Case((8, 8, 12), BC_SETS[bcname], seed=6)
    L = oracle.lib()
    u_raw, s_raw = case.random_state(with_ghost_fill=False)
    gu, gs = case.gmf(u_raw), case.gmf(s_raw)
    u, s = u_raw.copy(), s_raw.copy()
    L.vo_fill_boundary(u.ref, case.opm); L.vo_fill_boundary(s.ref, case.opm)
    L.vo_physbc(u.ref, 0, 0, 3, C.byref(case.obc), C.byref(case.prm))
    L.vo_physbc(s.ref, 0, 3, case.prm.nscal, C.byref(case.obc), C.byref(case.prm))
    gu.fill_boundary(); gs.fill_boundary()
    gu.physbc(0, 0, 3, case.bct); gs.physbc(0, 3, case.prm.nscal, case.bct)
    assert_bits(gu.to_numpy(), u.a, "physbc(u) " + bcname)
    assert_bits(gs.to_numpy(), s.a, "physbc(s) " + bcname)
    # face-centred and nodal fill_boundary
    for nodal in ((1, 0, 0), (0, 0, 1), (1, 1, 1)):
        f = case.ofab(1, 1, nodal)
        f.a[...] = case.rng.standard_normal(f.a.shape)
        for d in range(3):          # a nodal point on a periodic hi face IS the lo-face point
            if nodal[d] and case.pmask[d]:
                hi_sl, lo_sl = [slice(None)] * 4, [slice(None)] * 4
                hi_sl[d], lo_sl[d] = -2, 1
                f.a[tuple(hi_sl)] = f.a[tuple(lo_sl)]
        g = case.gmf(f)
        L.vo_fill_boundary(f.ref, case.opm)
        g.fill_boundary()
        assert_bits(g.to_numpy(), f.a, "fill_boundary nodal=%r %s" % (nodal, bcname))
    gp, ext = case.ofab(1, 3), case.ofab(1, 3)
    gp.a[...] = case.rng.standard_normal(gp.a.shape)
    ext.a[..., 2] = -9.8
    for dtold in (1.0e20, 1.0e-4):
        odt = L.vo_estdt(u.ref, s.ref, gp.ref, ext.ref, case.odx, C.c_double(dtold), C.byref(case.prm))
        gdt = adv.estdt(1, gu, gs, case.gmf(gp), case.gmf(ext), case.dx, dtold)
        assert gdt == odt, "estdt %r vs %r" % (gdt, odt)
    case.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(24, 22, 26), (70, 14, 33)])
def test_godunov_marching_equals_face_centred(gpu, shape):
    """three forms of the Godunov stages agree bit for bit: the default (mkflux: stages B + C + D fused into one march per component,
    velpred: one march per stage), the unfused marches (VDN_GOD_FUSED=0) and the face-centred one-thread-per-cell kernels
    (VDN_GODUNOV_PLAIN=1, there also with the per-cell slopes kernel instead of the marching one); the switches are read at the first launch of a
    process, hence the child processes.  The second shape spans
    two x-tiles, three y-tiles and several k-chunks of the fused march; the faces carry all four boundary rules."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        from varden_amd import advance as adv, boxlib as bl, capi
        n = %d
        bl.initialize(capi.default_params(), 0, 1, 0)
        lo, hi = (0, 0, 0), (%d - 1, %d - 1, %d - 1)
        mla = bl.MLLayout([(lo, hi)], [[(lo, hi)]])
        bct = bl.BCTower(mla, [[bl.INLET, bl.OUTLET], [bl.SLIP_WALL, bl.NO_SLIP_WALL], [bl.NO_SLIP_WALL, bl.OUTLET]])
        rng = np.random.default_rng(7)
        def mf(nc, ng, nodal=None):
            m = bl.MultiFab(mla, 0, nc, ng, nodal); m.from_numpy(rng.standard_normal(m.shape(0))); return m
        u, s = mf(3, 3), mf(2, 3)
        f3, f2, rhs = mf(3, 1), mf(2, 1), mf(1, 1)
        nd = [tuple(1 if t == d else 0 for t in range(3)) for d in range(3)]
        umac = [mf(1, 1, nd[d]) for d in range(3)]
        ue = [mf(3, 0, nd[d]) for d in range(3)]; uf = [mf(3, 0, nd[d]) for d in range(3)]
        se = [mf(2, 0, nd[d]) for d in range(3)]; sf = [mf(2, 0, nd[d]) for d in range(3)]
        dx = [1.0 / n] * 3
        adv.velpred(u, umac, f3, dx, 0.3 / n, bct)
        for m in umac: m.fill_boundary()
        adv.mkflux(u, ue, uf, umac, f3, rhs, dx, 0.3 / n, bct, True, [0, 0, 0])
        adv.mkflux(s, se, sf, umac, f2, rhs, dx, 0.3 / n, bct, False, [1, 0])
        h = hashlib.sha256()
        for m in umac + ue + se + [sf[d] for d in range(3)]:
            h.update(np.ascontiguousarray(m.to_numpy()).tobytes())
        print("HASH", h.hexdigest())
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), max(shape), shape[0], shape[1], shape[2]))
    out = []
    for extra in ({}, {"VDN_GOD_FUSED": "0"}, {"VDN_GODUNOV_PLAIN": "1", "VDN_SLOPES_MARCH": "0"}, {"VDN_FUSED_KCHUNKS": "5"}):
        env = dict(os.environ)
        for k in ("VDN_GODUNOV_PLAIN", "VDN_GOD_FUSED", "VDN_FUSED_KCHUNKS", "VDN_SLOPES_MARCH"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][0])
    assert out[0] == out[1] == out[2] == out[3], out


def _pair_run(tmp_path, tag, decomp, pair):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, VDN_GSRB_PAIR=str(pair))
    subprocess.check_call([sys.executable, os.path.join(root, "tests", "_pair_worker.py")] + [str(d) for d in decomp] + [out], env=env, cwd=root, timeout=300)
    return np.load(out)


@pytest.mark.gpu
def test_paired_colour_pass_equals_cell_per_thread_pass(gpu, tmp_path):
    """the finest level of the MAC multigrid: kk_cc_gsrb_rho_pair / kk_cc_residual_rho_pair (a thread owns a 2 x 2 block, 16-byte loads, lane
    exchange) against kk_cc_gsrb_rho / kk_cc_residual_rho (one cell per thread), and the stored-coefficient pair pass kk_cc_gsrb_pair of the
    128^3 level and of the three viscous solves (alpha = rho) against kk_cc_gsrb -- same arithmetic, same bits, on one 256 x 128 x 128 box and on two 128^3 boxes (colour by
    global index, halo cells from the neighbour box); and two boxes against one box to the usual 1e-9"""
    one_p, one_c = _pair_run(tmp_path, "one_p", (1, 1, 1), 1), _pair_run(tmp_path, "one_c", (1, 1, 1), 0)
    two_p, two_c = _pair_run(tmp_path, "two_p", (2, 1, 1), 1), _pair_run(tmp_path, "two_c", (2, 1, 1), 0)
    for a, b, what in ((one_p, one_c, "one box"), (two_p, two_c, "two boxes")):
        assert int(a["cyc"]) == int(b["cyc"]) and int(a["cyc"]) > 3
        assert_bits(a["u"], b["u"], "paired vs cell-per-thread colour pass, " + what)
        assert_bits(a["s"], b["s"], "paired vs cell-per-thread colour pass, " + what)
    assert np.abs(one_p["u"] - two_p["u"]).max() <= 1e-9 * np.abs(one_p["u"]).max()
