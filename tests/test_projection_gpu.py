"""GPU parity of the two projection solves (MAC cell-centred multigrid, HG nodal multigrid) and of the
projection drivers against the CPU oracle.  Both sides run the same algorithm in the same expression
order; the only order-dependent reduction is a max-norm, so the results are expected to be identical.
Tolerance written here: bit-exact for the smoother; 1e-11 relative (L-inf) for the converged solves
(tolerance-limited quantities: the reference itself stops at 1e-10 / 1e-12 relative residual)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import BC_SETS, Case, assert_bits

pytestmark = pytest.mark.gpu


def face_fabs(case, ng, nc, val=0.0):
    return [case.ofab(ng, nc, tuple(1 if t == d else 0 for t in range(3)), val) for d in range(3)]


def mac_problem(case):
    """rho bubble-like, beta from mk_mac_coeffs, a compatible rhs"""
    L = __import__("oracle.voracle", fromlist=["x"]).lib()
    _, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    beta = face_fabs(case, 0, 1)
    from oracle import voracle as vo
    L.vo_mk_mac_coeffs(s.ref, vo.fab_ptr_array(beta))
    rh = case.ofab(0, 1)
    rh.a[...] = case.rng.standard_normal(rh.a.shape)
    ell = vo.ellbc_of(case.obc)
    if all(ell[d][sd] != 1 for d in range(3) for sd in range(2)):      # no Dirichlet face: make it solvable
        rh.a[...] -= rh.a.mean()
    return s, beta, rh, ell


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout", "mixed"])
@pytest.mark.parametrize("n", [(16, 8, 12), (72, 36, 44)])
def test_cc_smoother_bits(gpu, oracle, bcname, n):
    from varden_amd import advance as adv
    case = Case(n, BC_SETS[bcname], seed=11, iso=True)
    s, beta, rh, ell = mac_problem(case)
    ophi = case.ofab(1, 1)
    oracle.lib().vo_cc_smooth(rh.ref, ophi.ref, oracle.fab_ptr_array(beta), case.odx, ell, 3)
    gphi = case.gmf(case.ofab(1, 1))
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    adv.cc_smooth(case.gmf(rh), gphi, [case.gmf(b) for b in beta], case.dx, bc, 3)
    g = gphi.to_numpy()
    assert_bits(g[1:-1, 1:-1, 1:-1], ophi.a[1:-1, 1:-1, 1:-1], "3 RB-GS sweeps " + bcname)
    case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout", "mixed"])
@pytest.mark.parametrize("n", [(16, 16, 16), (32, 16, 8), (12, 20, 24), (128, 16, 32)])      # 128 wide: the 2 x 2 pair passes on a flat box
def test_cc_solve(gpu, oracle, bcname, n):
    from varden_amd import advance as adv
    case = Case(n, BC_SETS[bcname], seed=12, iso=True)
    s, beta, rh, ell = mac_problem(case)
    P = case.prm
    ophi = case.ofab(1, 1)
    st = oracle.CMgStat()
    rc = oracle.lib().vo_cc_solve(rh.ref, ophi.ref, oracle.fab_ptr_array(beta), case.odx, ell, C.c_double(1e-10), C.c_double(-1.0), 100,
                                  P.mg_nu1, P.mg_nu2, P.mg_nub, P.mac_fmg, C.byref(st))
    assert rc == 0, "oracle MG did not converge"
    gphi = case.gmf(case.ofab(1, 1))
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    cyc, r0, r = adv.cc_solve(case.gmf(rh), gphi, [case.gmf(b) for b in beta], case.dx, bc, 1e-10)
    assert cyc == st.cycles and r0 == st.res0
    g = gphi.to_numpy()[1:-1, 1:-1, 1:-1, 0]
    o = ophi.a[1:-1, 1:-1, 1:-1, 0]
    scale = np.abs(o - o.mean()).max()
    err = float(np.abs(g - o).max())
    assert err <= 1e-11 * scale, "phi differs: %.3e (scale %.3e)" % (err, scale)
    assert r <= 1e-10 * r0
    case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout", "mixed"])
def test_macproject(gpu, oracle, bcname):
    """whole MAC projection: div(umac) = 0 afterwards (known-answer, macproject.f90:209-221) and parity"""
    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    case = Case((16, 16, 16), BC_SETS[bcname], seed=13, iso=True)
    L = oracle.lib()
    _, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    L.vo_fill_boundary(s.ref, case.opm)
    L.vo_physbc(s.ref, 0, 3, case.prm.nscal, C.byref(case.obc), C.byref(case.prm))
    # umac from velpred of a random velocity (so that boundary faces carry the bc values)
    u, _ = case.random_state()
    force = case.ofab(1, 3)
    oum = face_fabs(case, 1, 1, 1.0e20)
    L.vo_velpred(u.ref, oracle.fab_ptr_array(oum), force.ref, case.odx, C.c_double(0.2 * min(case.dx)), C.byref(case.obc), C.byref(case.prm))
    for f in oum:
        L.vo_fill_boundary(f.ref, case.opm)
    mac_rhs = case.ofab(1, 1)
    gum = [case.gmf(f) for f in oum]
    st = oracle.CMgStat()
    L.vo_macproject(oracle.fab_ptr_array(oum), s.ref, mac_rhs.ref, case.odx, C.byref(case.obc), case.opm, C.byref(case.prm), C.byref(st))
    adv.macproject(case.mla, [gum], [case.gmf(s)], [case.gmf(mac_rhs)], [case.dx], case.bct, case.obc.press_comp + 1)
    cyc, r0, r = adv.last_solver_stats("mac")
    assert cyc == st.cycles
    for d in range(3):
        g, o = gum[d].to_numpy(), oum[d].a
        scale = np.abs(o[1:-1, 1:-1, 1:-1]).max()
        assert np.abs(g - o)[1:-1, 1:-1, 1:-1].max() <= 1e-11 * scale
    # divergence-free to the solver tolerance
    div = sum((np.diff(gum[d].to_numpy()[1:-1, 1:-1, 1:-1, 0], axis=d)[tuple(slice(0, case.n[t]) for t in range(3))]) / case.dx[d] for d in range(3))
    assert np.abs(div).max() <= 1e-9 * r0, "max |div umac| = %.3e (initial %.3e)" % (np.abs(div).max(), r0)
    case.close()


@pytest.mark.parametrize("bcname,n", [("walls", (132, 36, 40)), ("inout", (132, 36, 40)), ("zout", (132, 36, 40)), ("inout", (260, 20, 24))])
def test_split_colour_level_matches_the_oracle_and_the_interleaved_level(gpu, oracle, bcname, n):
    """round 5: macproject's one-box solve keeps its finest level BY COLOUR from 2^23 cells up (kk_cc_gsrb_rho_split, kk_cc_residual_rho_split_rst; the
    256^3 tests run it).  Here on 132 x 36 x 40 cells (VDN_MAC_SPLIT_MIN=0; 33 lane pairs: the clamped tail of a wave, one-sided y / z extents): the
    projected velocities against the oracle's in the worker, and the same bits (a) split, the passes and the residual time-skewed over slabs of 7 planes
    (cc_split_run; 40 planes: six slabs, the last a sliver), the second colour walking its planes downwards,
    (b) split passes only, residual on the level array, both colours upwards, (c) interleaved.  260 cells: two waves per row, the second with one active lane pair
    (the lane that ends a wave inside the row reads its neighbour from memory).  tests/_split_worker.py."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, form = [], []
    for extra in ({"VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_SLAB": "7"}, {"VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_SPLIT": "2", "VDN_MAC_KFLIP": "0"}, {"VDN_MAC_SPLIT": "0"}):
        env = dict(os.environ)
        for k in ("VDN_MAC_SPLIT", "VDN_MAC_SPLIT_MIN", "VDN_MAC_KFLIP", "VDN_MAC_SLAB"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_split_worker.py"), bcname] + [str(v) for v in n], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
        form.append([ln for ln in r.stdout.splitlines() if ln.startswith("FORM")][0])
    assert form == ["FORM 1", "FORM 2", "FORM 0"], form             # (vdn_last_mac_level_form: the three runs took the three paths)
    assert out[0] == out[1] == out[2], (bcname, out)


@pytest.mark.parametrize("bcname,n,nb", [("periodic", (132, 36, 40), (1, 1, 1)), ("periodicyz", (132, 36, 40), (1, 1, 1)), ("walls", (264, 128, 128), (2, 1, 1)),
                                         ("periodicx", (264, 128, 128), (2, 1, 1)), ("inout", (132, 256, 128), (1, 2, 1)), ("zout", (132, 128, 256), (1, 1, 2))])
def test_split_colour_level_with_a_halo(gpu, oracle, bcname, n, nb):
    """round 6: the level by colour also where a ghost exchange runs between the passes -- periodic faces of one box (the box is its own neighbour), several boxes
    (two along x: the ghost ENTRY -1 / nh of a row; two along y / along z: ghost rows / planes; x periodic with two boxes: each the other's neighbour on both sides).  The exchange
    runs on the split arrays (cc_split_plan: phi of one colour in the index space halved along x), the correction of the coarse level rides in the first sweep with the
    coarse level's ghost cells exchanged, residual + restriction per box.  Against the oracle in the first run; the same bits (a) split, (b) split with the exchange on
    the halo stream next to the interior cells and the shell kernel behind it (VDN_OVERLAP=1), (c) split through the packed per-peer buffers (VDN_FORCE_PACKED=1: the
    path to another rank), (d) interleaved (round 5's form of these levels).  Boxes of 132 x 128 x 128 cells: the smallest that keep a second distributed level."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, form = [], []
    variants = ({"VDN_MAC_SPLIT_MIN": "0"}, {"VDN_MAC_SPLIT_MIN": "0", "VDN_OVERLAP": "1", "VDN_WORKER_ORACLE": "0"},
                {"VDN_MAC_SPLIT_MIN": "0", "VDN_FORCE_PACKED": "1", "VDN_WORKER_ORACLE": "0"}, {"VDN_MAC_SPLIT": "0", "VDN_WORKER_ORACLE": "0"})
    if nb[0] == 1 and nb != (1, 1, 1):          # (the y / z decompositions: split with the overlap path against interleaved -- the packed path's descriptors do not depend on the direction)
        variants = (variants[0], variants[1], variants[3])
    for extra in variants:
        env = dict(os.environ)
        for k in ("VDN_MAC_SPLIT", "VDN_MAC_SPLIT_MIN", "VDN_MAC_KFLIP", "VDN_MAC_SLAB", "VDN_OVERLAP", "VDN_FORCE_PACKED", "VDN_MAC_SPLIT_HALO", "VDN_MG_AGGLOM"):
            env.pop(k, None)
        env.update(extra)
        env["VDN_MG_AGGLOM"] = "64"          # (boxes of 128 cells on ONE rank are gathered right below the finest level by default, mg_agglom: keep the second distributed level the split form asks for)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_split_worker.py"), bcname] + [str(v) for v in n + nb], env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
        form.append([ln for ln in r.stdout.splitlines() if ln.startswith("FORM")][0])
    assert form == ["FORM 1"] * (len(variants) - 1) + ["FORM 0"], form
    assert len(set(out)) == 1, (bcname, out)


def test_blown_up_field_fails_loudly(gpu, oracle):
    """the norms are NaN-propagating (a NaN residual used to read as 0 = converged) and a failed solve fails the call,
    as FBoxLib's solvers abort on max_iter; abort_on_max_iter = 0 restores report-and-continue"""
    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    case = Case((16, 16, 16), BC_SETS["walls"], seed=3, iso=True)
    u, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    um = face_fabs(case, 1, 1, 0.0)
    um[0].a[8, 8, 8, 0] = np.nan
    mac_rhs = case.ofab(1, 1)
    gum = [case.gmf(f) for f in um]
    with pytest.raises(Exception, match="non-finite"):
        adv.macproject(case.mla, [gum], [case.gmf(s)], [case.gmf(mac_rhs)], [case.dx], case.bct, case.obc.press_comp + 1)
    u.a[5, 5, 5, 1] = np.inf
    gp, ext = case.ofab(1, 3), case.ofab(1, 3)
    with pytest.raises(Exception, match="non-finite"):
        adv.estdt(1, case.gmf(u), case.gmf(s), case.gmf(gp), case.gmf(ext), case.dx, 1.0)
    # max_iter reached: one cycle cannot reach 1e-10
    s2, beta, rh, ell = mac_problem(case)
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    with pytest.raises(Exception, match="did not converge"):
        adv.cc_solve(case.gmf(rh), case.gmf(case.ofab(1, 1)), [case.gmf(b) for b in beta], case.dx, bc, 1e-10, max_iter=1)
    case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout", "mixed"])
@pytest.mark.parametrize("n", [(16, 16, 16), (8, 16, 32), (128, 16, 32)])      # 128 wide: the paired march and the fused residual + restriction (kk_nd_march_pair_rst) on a flat box
def test_nd_solve(gpu, oracle, bcname, n):
    from varden_amd import advance as adv
    case = Case(n, BC_SETS[bcname], seed=14, iso=True)
    L = oracle.lib()
    P = case.prm
    u, s = case.random_state()
    # zero wall ghosts like create_uvec does, so the system is compatible
    rhohalf = case.ofab(1, 1)
    rhohalf.a[...] = np.abs(s.a[2:-2, 2:-2, 2:-2, :1]) + 0.5
    gpz = case.ofab(1, 3)
    L.vo_create_uvec(u.ref, u.ref, rhohalf.ref, gpz.ref, C.c_double(1.0), C.byref(case.obc), 1)
    L.vo_fill_boundary(u.ref, case.opm)
    coeffs = case.ofab(1, 1)
    coeffs.a[1:-1, 1:-1, 1:-1, 0] = 1.0 / rhohalf.a[1:-1, 1:-1, 1:-1, 0]
    L.vo_fill_boundary(coeffs.ref, case.opm)
    ell = oracle.ellbc_of(case.obc)
    nodal = (1, 1, 1)
    orh, ophi = case.ofab(1, 1, nodal), case.ofab(1, 1, nodal)
    st = oracle.CMgStat()
    rc = L.vo_nd_solve(orh.ref, ophi.ref, coeffs.ref, u.ref, case.odx, ell, case.opm, C.c_double(1e-11), C.c_double(-1.0), 100,
                       P.hg_nu1, P.hg_nu2, P.hg_nub, C.c_double(P.hg_omega), P.hg_fmg, (C.c_double * 2)(P.hg_omega_pre1, P.hg_omega_pre2), C.byref(st))
    assert rc == 0, "oracle nodal MG did not converge (%d cycles, %g / %g)" % (st.cycles, st.res, st.res0)
    grh, gphi = case.gmf(case.ofab(1, 1, nodal)), case.gmf(case.ofab(1, 1, nodal))
    bc = [[ell[d][sd] for sd in range(2)] for d in range(3)]
    cyc, r0, r = adv.nd_solve(grh, gphi, case.gmf(coeffs), case.gmf(u), case.dx, bc, 1e-11)
    assert cyc == st.cycles and r0 == st.res0
    assert_bits(grh.to_numpy()[1:-1, 1:-1, 1:-1], orh.a[1:-1, 1:-1, 1:-1], "nodal divergence " + bcname)
    g, o = gphi.to_numpy()[1:-1, 1:-1, 1:-1, 0], ophi.a[1:-1, 1:-1, 1:-1, 0]
    scale = np.abs(o - o.mean()).max()
    err = float(np.abs(g - o).max())
    assert err <= 1e-11 * scale, "phi differs: %.3e (scale %.3e)" % (err, scale)
    case.close()


@pytest.mark.parametrize("bcname", ["walls", "periodic", "inout"])
@pytest.mark.parametrize("proj_type", [1, 3, 4])
def test_hgproject(gpu, oracle, bcname, proj_type):
    from varden_amd import advance as adv
    case = Case((16, 16, 16), BC_SETS[bcname], seed=15, iso=True)
    L = oracle.lib()
    uold, s = case.random_state()
    unew, _ = case.random_state()
    rhohalf = case.ofab(1, 1)
    rhohalf.a[...] = np.abs(s.a[2:-2, 2:-2, 2:-2, :1]) + 0.5
    gp, p = case.ofab(1, 3), case.ofab(1, 1, (1, 1, 1))
    gp.a[1:-1, 1:-1, 1:-1] = case.rng.standard_normal(gp.a[1:-1, 1:-1, 1:-1].shape)
    L.vo_fill_boundary(gp.ref, case.opm)
    p.a[...] = case.rng.standard_normal(p.a.shape)
    dt = 0.01
    g_un, g_uo, g_rh, g_p, g_gp = case.gmf(unew), case.gmf(uold), case.gmf(rhohalf), case.gmf(p), case.gmf(gp)
    st = oracle.CMgStat()
    L.vo_hgproject(proj_type, unew.ref, uold.ref, rhohalf.ref, p.ref, gp.ref, case.odx, C.c_double(dt), C.byref(case.obc), case.opm,
                   C.byref(case.prm), C.byref(st))
    adv.hgproject(proj_type, case.mla, [g_un], [g_uo], [g_rh], [g_p], [g_gp], [case.dx], dt, case.bct, case.obc.press_comp + 1)
    cyc, r0, r = adv.last_solver_stats("hg")
    assert cyc == st.cycles, "cycles %d vs %d" % (cyc, st.cycles)
    v = (slice(3, -3),) * 3
    for name, g, o, sl in (("unew", g_un.to_numpy(), unew.a, v), ("gp", g_gp.to_numpy(), gp.a, (slice(1, -1),) * 3),
                           ("p", g_p.to_numpy(), p.a, (slice(1, -1),) * 3)):
        scale = max(np.abs(o[sl]).max(), 1e-300)
        err = np.abs(g[sl] - o[sl]).max()
        assert err <= 1e-10 * scale, "%s differs after hgproject(%d): %.3e (scale %.3e)" % (name, proj_type, err, scale)
    case.close()


def test_multigrid_launch_variants_agree_bit_for_bit(gpu):
    """the launch-saving forms of the V-cycles change no value: a 128^3 step (the finest MAC level takes the paired density pass only from
    128 cells up) with (a) the defaults -- prolongation added inside the first post-smoothing sweep (kk_cc_gsrb_rho_pair_t), restriction inside the
    residual pass (kk_cc_residual_rho_pair_rst; nodal: the x- and z-sums of the full weighting inside the residual march, kk_nd_march_pair_rst + kk_nd_rst_y), the 16^3 .. 64^3 levels of the cell-centred solver as one LDS-tiled launch down and one up (kk_cc_lds_down /
    kk_cc_lds_up), the levels of
    at most 9^3 nodes / 8^3 cells in one single-workgroup launch (kk_*_tailcycle), V-cycles replayed as hipGraphs -- against (b) the
    plain sequence of launches, with the rh / phi / coeffs / beta multifabs of hgproject and macproject as the reference has them (VDN_HG_FAST=0, VDN_MAC_FAST=0), whole-array zero fills
    (VDN_ND_LEAN=0), every forcing term computed where the reference computes it (VDN_NO_FORCE_REUSE=1) and the Godunov marches dividing by dx where the
    default scales by 1 / dx on these power-of-two grids (VDN_GOD_P2=0); and a viscous 64^3 run the same way (the alpha form of the cell-centred kernels: three visc_solves per step).
    The switches are read once per process, hence the child processes."""
    import hashlib, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        from varden_amd import driver
        from varden_amd.capi import default_params
        n, visc = int(sys.argv[1]), float(sys.argv[2])
        G = driver.VardenAMR(n, [], [[15, 15]] * 3, params=default_params(visc_coef=visc, diff_coef=visc), init_iter=1, do_initial_projection=1)
        for _ in range(2):
            G.step()
        h = hashlib.sha256()
        for m in (G.uold[0], G.sold[0], G.p[0], G.gp[0]):
            h.update(np.ascontiguousarray(m.to_numpy()).tobytes())
        print("HASH", h.hexdigest(), G.dt)
    """ % root)
    switches = ("VDN_MG_PROLONG_FUSED", "VDN_MG_RESTRICT_FUSED", "VDN_MG_TAILCYCLE", "VDN_MG_LDS", "VDN_NO_GRAPHS", "VDN_HG_FAST", "VDN_MAC_FAST", "VDN_ND_LEAN", "VDN_NO_FORCE_REUSE", "VDN_GOD_UPDATE", "VDN_GOD_P2", "VDN_ND_RESTRICT_FUSED", "VDN_GOD_1B", "VDN_GOD_NARROW", "VDN_MAC_SPLIT", "VDN_MAC_SPLIT_MIN", "VDN_MAC_KFLIP", "VDN_ND_REV", "VDN_MAC_SLAB",
                "VDN_SLOPES_Y", "VDN_MAC_UMAX", "VDN_ND_PAIR", "VDN_MAC_STORED_BETA", "VDN_NO_SLOPE_CACHE", "VDN_CC_HALO_FACES", "VDN_GOD_SLAB_BC", "VDN_GODUNOV_BATCH")
    for n, visc in ((128, 0.0), (64, 0.01)):
        out = []
        # third run (round 5): the defaults with the fused mkflux + update march in its round-4 form -- three workgroup barriers per plane, the
        # remainder tile column in full 64-lane tiles -- against one barrier and narrow segments
        # round 5 also: the first two runs keep the finest MAC level by colour (VDN_MAC_SPLIT_MIN=0: from any size), the third interleaved; the second and third
        # walk every colour pass / nodal march in the same order (VDN_MAC_KFLIP=0, VDN_ND_REV=0); the first in two plane slabs (cc_split_run), the second in whole-level launches
        for extra in ({"VDN_MAC_SPLIT_MIN": "0"}, {"VDN_GOD_1B": "0", "VDN_GOD_NARROW": "0", "VDN_SLOPES_Y": "0", "VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_KFLIP": "0", "VDN_ND_REV": "0", "VDN_MAC_SLAB": "0", "VDN_NO_GRAPHS": "1"},
                      {"VDN_MAC_KFLIP": "0", "VDN_ND_REV": "0", "VDN_MAC_SPLIT": "0", "VDN_MAC_UMAX": "0", "VDN_ND_PAIR": "0", "VDN_MAC_STORED_BETA": "1", "VDN_NO_SLOPE_CACHE": "1", "VDN_CC_HALO_FACES": "0",
                       "VDN_GOD_SLAB_BC": "0", "VDN_GODUNOV_BATCH": "1", "VDN_MG_PROLONG_FUSED": "0", "VDN_MG_RESTRICT_FUSED": "0", "VDN_MG_TAILCYCLE": "0", "VDN_MG_LDS": "0", "VDN_NO_GRAPHS": "1",
                          "VDN_HG_FAST": "0", "VDN_MAC_FAST": "0", "VDN_ND_LEAN": "0", "VDN_NO_FORCE_REUSE": "1", "VDN_GOD_UPDATE": "0", "VDN_GOD_P2": "0", "VDN_ND_RESTRICT_FUSED": "0"}):
            env = dict(os.environ)
            for k in switches:
                env.pop(k, None)
            env.update(extra)
            r = subprocess.run([sys.executable, "-c", code, str(n), str(visc)], env=env, capture_output=True, text=True, timeout=600, cwd=root)
            assert r.returncode == 0, r.stderr[-2000:]
            out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
        assert out[0] == out[1] == out[2], (n, visc, out)
