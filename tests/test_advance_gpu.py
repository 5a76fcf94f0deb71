"""End-to-end parity of advance_timestep (through the C-ABI, driven exactly like reference
src/varden.f90 drives it) against the CPU oracle on the bubble problem of exec/test/inputs_bubble_3d
(inviscid), plus size-independent properties at a larger size.
Tolerance: rel L-inf 1e-9 on u and rho after several steps; in practice the two paths
agree to ~1e-13 because they run the same algorithm in the same expression order."""
import numpy as np
import pytest

from tests.util import WALLS, PER, INOUT, params_for

pytestmark = pytest.mark.gpu


def run_pair(n, phys, nsteps, prob_type=1, **kw):
    from oracle import voracle as vo
    from varden_amd import driver
    prm_o, prm_g = params_for(phys, cflfac=0.9), params_for(phys, cflfac=0.9)
    O = vo.Sim(n, phys, prm_o, prob_type=prob_type, init_shrink=0.1, init_iter=1, **kw)
    ou, os_ = vo.Fab((0, 0, 0), tuple(x - 1 for x in O.n), 3, 3), vo.Fab((0, 0, 0), tuple(x - 1 for x in O.n), 3, 2)
    vo.lib().vo_initdata(ou.ref, os_.ref, O.dx, prob_type)
    G = driver.Varden(n, phys, prm_g, prob_type=prob_type, init_shrink=0.1, init_iter=1, u0=ou.a, s0=os_.a, **kw)
    assert G.dt == O.dt
    for _ in range(nsteps):
        O.step(); G.step()
        assert G.dt == O.dt, "dt diverged: %r vs %r" % (G.dt, O.dt)
    return O, G


@pytest.mark.parametrize("name,phys,prob", [("bubble-walls", WALLS, 1), ("bubble-periodic", PER, 1), ("blob-inout", INOUT, 2)])
def test_advance_parity_small(gpu, name, phys, prob):
    O, G = run_pair(16, phys, 3, prob_type=prob)
    g = 3
    for nm, gm, om in (("u", G.unew[0], O.unew), ("s", G.snew[0], O.snew)):
        a, b = gm.to_numpy()[g:-g, g:-g, g:-g], om.valid()
        scale = max(np.abs(b).max(), 1e-300)
        assert np.abs(a - b).max() <= 1e-9 * scale, "%s: %s differs by %.3e (scale %.3e)" % (name, nm, np.abs(a - b).max(), scale)
    a, b = G.p[0].to_numpy()[1:-1, 1:-1, 1:-1], O.p.valid()
    assert np.abs((a - a.mean()) - (b - b.mean())).max() <= 1e-6 * max(np.abs(b - b.mean()).max(), 1e-300)
    G.close()


PERX = [[-1, -1], [15, 15], [15, 15]]            # periodic along x, no-slip walls elsewhere


@pytest.mark.parametrize("n,name,phys", [(64, "walls", WALLS), (64, "periodic-x", PERX), (128, "walls", WALLS), (128, "periodic-x", PERX), (256, "walls", WALLS)])
def test_advance_parity_large(gpu, n, name, phys):
    """the full step against the oracle at the sizes where the production launch forms run INSIDE an oracle-checked step: at
    64^3 and 128^3 the multigrids have 5-7 levels -- the XCD-ordered 2 x 2 pair colour pass, the LDS-tiled 16^3-64^3 levels, the single-workgroup tail
    cycles, the fused residual + restriction marches, the nested-iteration starts and mg_predict -- and the Godunov marches run several tiles and
    chunks per plane with the power-of-two spacing form.  Start-up sequence (initial projection, one pressure iteration) + 2 steps; u, rho, tracer
    to 1e-9, the pressure to 1e-6, dt bit for bit, equal V-cycle counts of both projections in every step.  Round 5: 256^3, BASELINE.json configs[1] itself, one
    step -- the MAC solve's finest level stored by colour, its passes time-skewed over plane slabs (vdn_last_mac_level_form says so), eight multigrid levels."""
    from oracle import voracle as vo
    from varden_amd import advance as adv
    from varden_amd import driver
    O = vo.Sim(n, phys, params_for(phys, cflfac=0.9), prob_type=1, init_shrink=0.1, init_iter=1)
    G = driver.Varden(n, phys, params_for(phys, cflfac=0.9), prob_type=1, init_shrink=0.1, init_iter=1)
    assert G.initial_projection_stat[0] == O.initial_projection_stat[0], "initial projection: V-cycle counts differ"
    assert G.dt == O.dt
    g = 3
    for step in range(1 if n == 256 else 2):
        O.step(); G.step()
        assert G.dt == O.dt, "dt diverged at step %d: %r vs %r" % (step, G.dt, O.dt)
        cg = (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
        co = (O.mgstat[0].cycles, O.mgstat[1].cycles)
        assert cg == co, "%d^3 %s step %d: V-cycle counts (MAC, HG) %r on the GPU, %r in the oracle" % (n, name, step, cg, co)
        if n == 256:
            from varden_amd import capi
            assert capi.load().vdn_last_mac_level_form() == 1, "the 256^3 MAC solve should have kept its finest level by colour"
        for nm, gm, om in (("u", G.unew[0], O.unew), ("s", G.snew[0], O.snew)):
            a, b = gm.to_numpy()[g:-g, g:-g, g:-g], om.valid()
            scale = max(float(np.abs(b).max()), 1e-300)
            err = float(np.abs(a - b).max())
            assert err <= 1e-9 * scale, "%d^3 %s step %d: %s differs by %.3e (scale %.3e)" % (n, name, step, nm, err, scale)
        a, b = G.p[0].to_numpy()[1:-1, 1:-1, 1:-1], O.p.valid()
        assert np.abs((a - a.mean()) - (b - b.mean())).max() <= 1e-6 * max(np.abs(b - b.mean()).max(), 1e-300)
    G.close()


def test_advance_properties_64(gpu):
    """size-independent checks at 64^3 (no oracle run): symmetry of the bubble about x=y=1/2, conservation of
    mass to round-off with the conservative density update (update.f90:250-253) under wall bcs (zero boundary
    flux), finite fields, solver convergence"""
    from varden_amd import advance as adv
    from varden_amd import driver
    G = driver.Varden(64, WALLS, params_for(WALLS, cflfac=0.9), init_shrink=0.1, init_iter=1)
    m0 = G.sold[0].to_numpy()[3:-3, 3:-3, 3:-3, 0].sum()
    for _ in range(3):
        G.step()
        assert adv.last_solver_stats("mac")[0] < 30 and adv.last_solver_stats("hg")[0] < 40
    u = G.unew[0].to_numpy()[3:-3, 3:-3, 3:-3]
    s = G.snew[0].to_numpy()[3:-3, 3:-3, 3:-3]
    assert np.isfinite(u).all() and np.isfinite(s).all()
    assert abs(s[..., 0].sum() - m0) <= 1e-10 * m0
    # mirror symmetry x <-> 1-x: u odd, v,w,rho even (the scheme is not bitwise symmetric: upwind ties)
    assert np.abs(u[..., 0] + u[::-1, :, :, 0]).max() <= 1e-8 * np.abs(u).max()
    assert np.abs(u[..., 2] - u[::-1, :, :, 2]).max() <= 1e-8 * np.abs(u).max()
    assert np.abs(s[..., 0] - s[:, ::-1, :, 0]).max() <= 1e-8 * np.abs(s).max()
    assert u[..., 2].max() > 0 or u[..., 2].min() < 0     # gravity moved something
    G.close()


@pytest.mark.parametrize("name,phys,prob,dtype", [("bubble-walls-CN", WALLS, 1, 1), ("blob-inout-CN", INOUT, 2, 1), ("bubble-periodic-BE", PER, 1, 2)])
def test_advance_parity_viscous(gpu, name, phys, prob, dtype):
    """the shipped inputs all have visc_coef > 0 (exec/test/inputs_bubble_3d:33): explicit diffusive term + implicit
    Crank-Nicolson / backward-Euler viscous and scalar-diffusion solves (viscsolve.f90), against the oracle"""
    from oracle import voracle as vo
    from varden_amd import driver
    n = 16
    kw = dict(cflfac=0.9, visc_coef=0.01, diff_coef=0.005, diffusion_type=dtype)
    O = vo.Sim(n, phys, params_for(phys, **kw), prob_type=prob, init_shrink=0.1, init_iter=1)
    ou, os_ = vo.Fab((0, 0, 0), (n - 1,) * 3, 3, 3), vo.Fab((0, 0, 0), (n - 1,) * 3, 3, 2)
    vo.lib().vo_initdata(ou.ref, os_.ref, O.dx, prob)
    G = driver.Varden(n, phys, params_for(phys, **kw), prob_type=prob, init_shrink=0.1, init_iter=1, u0=ou.a, s0=os_.a)
    for _ in range(3):
        O.step(); G.step()
        assert G.dt == O.dt
    g = 3
    for nm, gm, om in (("u", G.unew[0], O.unew), ("s", G.snew[0], O.snew)):
        a_, b_ = gm.to_numpy()[g:-g, g:-g, g:-g], om.valid()
        scale = max(float(np.abs(b_).max()), 1e-300)
        err = float(np.abs(a_ - b_).max())
        assert err <= 1e-9 * scale, "%s: %s differs by %.3e (scale %.3e)" % (name, nm, err, scale)
    G.close()


def test_long_run_stays_with_the_oracle(gpu):
    """80 viscous steps of the falling blob at 24^3 -- through the phase in which it reaches the floor and the density leaves its initial
    bounds: time and time step stay within 1e-12, the fields within 1e-10 (tools/long_vs_oracle.py: 300
    steps at 32^3, 1e-14 at step 100, 2e-8 at step 300)"""
    from oracle import voracle as vo
    from varden_amd import driver
    mk = lambda: params_for(WALLS, cflfac=0.9, visc_coef=0.001)   # noqa: E731
    kw = dict(prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=2)
    O = vo.Sim(24, WALLS, mk(), **kw)
    G = driver.Varden(24, WALLS, mk(), **kw)
    for _ in range(80):
        O.step(); G.step()
    assert abs(G.time - O.time) <= 1e-12 * O.time and abs(G.dt - O.dt) <= 1e-12 * O.dt
    s, so = G.snew[0].to_numpy()[3:-3, 3:-3, 3:-3], O.snew.valid()
    u, uo = G.unew[0].to_numpy()[3:-3, 3:-3, 3:-3], O.unew.valid()
    assert so[..., 0].max() < 5.0                                   # the blob (rho = 10) has been smeared over the floor
    assert np.abs(s - so).max() <= 1e-10 * np.abs(so).max() and np.abs(u - uo).max() <= 1e-10 * np.abs(uo).max()
    G.close()


def test_inviscid_run_through_the_impact_stays_with_the_oracle(gpu):
    """the bench's configuration (inviscid blob between walls, cflfac 0.9) at 64^3 for 100 steps: the blob reaches the floor near step 45, |u| jumps from 3 to 7
    and the density starts to leave its bounds (at 128^3 for 170 steps: tools/long_vs_oracle_inviscid.py, 1e-14).  Every step: dt to 1e-12, the
    V-cycle counts of both projections equal; every 20 steps u and rho within 1e-11 of the oracle's."""
    from oracle import voracle as vo
    from varden_amd import advance as adv
    from varden_amd import driver
    kw = dict(prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1)
    O = vo.Sim(64, WALLS, params_for(WALLS, cflfac=0.9), **kw)
    G = driver.Varden(64, WALLS, params_for(WALLS, cflfac=0.9), **kw)
    umax = 0.0
    for step in range(100):
        O.step(); G.step()
        assert abs(G.dt - O.dt) <= 1e-12 * O.dt, "dt diverged at step %d: %r vs %r" % (step, G.dt, O.dt)      # (the fields agree to ~1e-14, not to the bit: neither does their max norm)
        cg = (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0])
        co = (O.mgstat[0].cycles, O.mgstat[1].cycles)
        assert cg == co, "step %d: V-cycle counts (MAC, HG) %r on the GPU, %r in the oracle" % (step, cg, co)
        if step % 20 == 19:
            for nm, gm, om in (("u", G.unew[0], O.unew), ("s", G.snew[0], O.snew)):
                a, b = gm.to_numpy()[3:-3, 3:-3, 3:-3], om.valid()
                scale = float(np.abs(b).max())
                assert float(np.abs(a - b).max()) <= 1e-11 * scale, "step %d: %s differs by %.3e (scale %.3e)" % (step, nm, np.abs(a - b).max(), scale)
            umax = max(umax, float(np.abs(O.unew.valid()).max()))
    assert umax > 5.0, "the blob should have hit the floor (max |u| %.2f)" % umax
    G.close()


@pytest.mark.parametrize("phys", [WALLS, PER, INOUT], ids=["walls", "periodic", "inout"])
def test_handle_swap_equals_copy(gpu, phys):
    """`swap_state=True` (what bench.py runs: uold <- unew by exchanging the multifab handles instead of varden.f90:323-326's copy of the
    valid cells) gives the same run bit for bit: every ghost cell of uold / sold is refilled before the next step reads it.  Four steps on
    two boxes (an interior box face as well as physical ones), state, pressure and dt compared."""
    from varden_amd import driver
    runs = []
    for swap in (False, True):
        G = driver.Varden((32, 16, 16), phys, params_for(phys, cflfac=0.9), prob_type=1, prob_hi=(2.0, 1.0, 1.0), init_shrink=0.1, init_iter=1,
                          decomp=(2, 1, 1), swap_state=swap)
        dts = []
        for _ in range(4):
            G.step(); dts.append(G.dt)
        runs.append((dts, [G.gather_valid(m) for m in (G.uold[0], G.sold[0], G.gp[0])], G.p[0].to_numpy(0)))
        G.close()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(runs[0][2], runs[1][2])


def test_handle_swap_equals_copy_two_levels(gpu):
    from varden_amd import driver
    runs = []
    for swap in (False, True):
        G = driver.VardenAMR(16, [((8, 8, 8), (23, 23, 23))], WALLS, init_shrink=0.1, init_iter=1, swap_state=swap)
        dts = []
        for _ in range(3):
            G.step(); dts.append(G.dt)
        runs.append((dts, [m.to_numpy(0)[3:-3, 3:-3, 3:-3] for n in range(2) for m in (G.uold[n], G.sold[n])]))
        G.close()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(*[r[1] for r in runs]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bcname,n,diff", [("walls", (132, 36, 40), 0.0), ("slipz", (132, 36, 40), 0.002), ("periodicx", (132, 36, 40), 0.0), ("inout", (260, 20, 24), 0.0)])
def test_viscous_solves_by_colour(gpu, oracle, bcname, n, diff):
    """round 6: the three Crank-Nicolson velocity solves of a viscous step (and the tracer's with diff_coef > 0) keep their finest level BY COLOUR -- the face coefficients of
    (rho - div mu grad) are the constant mu (viscsolve.f90:57-60), so a colour pass reads phi, rhs and alpha = rho only (kk_cc_gsrb_rho_split<., VISC>), 20 B per cell of the level
    where the stored-coefficient pass moves 56; residual + restriction fused, the correction inside the first sweep, the slab schedule.  By default from 2^23 cells (at 256^3 the
    three solves were 24 of 51 ms per step); here VDN_MAC_SPLIT_MIN=0 on 132 x 36 x 40 cells: two steps against the oracle (u, rho, tracer to 1e-9, equal V-cycle counts of both
    projections, dt bit for bit) and the same state hash (a) by colour with slabs of 7 planes, (b) by colour, whole-level launches, (c) interleaved.  Dirichlet walls (no-slip: every
    component), mixed (slip walls: the normal component Dirichlet, the tangential ones Neumann -- the folding differs per component), periodic x (ghost entries exchanged), inflow /
    outflow (inhomogeneous Dirichlet data in the right-hand side), 260 cells (two waves per row)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, form = [], []
    for extra in ({"VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_SLAB": "7"}, {"VDN_MAC_SPLIT_MIN": "0", "VDN_MAC_SLAB": "0", "VDN_WORKER_ORACLE": "0"}, {"VDN_MAC_SPLIT": "0", "VDN_WORKER_ORACLE": "0"}):
        env = dict(os.environ)
        for k in ("VDN_MAC_SPLIT", "VDN_MAC_SPLIT_MIN", "VDN_MAC_KFLIP", "VDN_MAC_SLAB"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_visc_split_worker.py"), bcname] + [str(v) for v in n] + [str(diff)], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
        form.append([ln for ln in r.stdout.splitlines() if ln.startswith("FORM")][0])
    assert form[0].split()[1] == "1" and form[1].split()[1] == "1" and form[2].split()[1] == "0", form       # (the start-up's last cell-centred solve: a viscous one)
    assert out[0] == out[1] == out[2], (bcname, out)
