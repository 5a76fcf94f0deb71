"""plot and checkpoint files.  (1) the derived plot quantities of src/makevort.f90 on the device against
the CPU oracle, bit for bit, for every boundary family in 3-D and 2-D and on several boxes; (2) a plotfile written from a running
hierarchy holds exactly the device data; (3) a run restarted from a checkpoint continues bit for bit (single level, several boxes,
and a two-level hierarchy)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import BC_SETS, WALLS, Case, assert_bits, params_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bcname", list(BC_SETS))
def test_vorticity_magvel_3d(gpu, oracle, bcname):
    from varden_amd import advance as adv
    case = Case((16, 12, 8), BC_SETS[bcname], seed=11)
    L = oracle.lib()
    u, _ = case.random_state()
    ov = case.ofab(0, 2)
    L.vo_makevort(ov.ref, 1, u.ref, case.odx, C.byref(case.obc))
    L.vo_makemagvel(ov.ref, 0, u.ref)
    raw = u.copy()
    raw.a[...] = np.where(np.isfinite(raw.a), raw.a, 0.0)
    gu, gv = case.gmf(raw), case.gmf(case.ofab(0, 2))
    adv.make_magvel(gv, 0, gu)
    adv.make_vorticity(gv, 1, gu, case.dx, case.bct)          # fills the ghost cells of u itself (makevort.f90:34-38)
    assert_bits(gv.to_numpy(), ov.a, "magvel / vort " + bcname)
    assert ov.a[..., 1].min() >= 0.0 and ov.a[..., 1].max() > 1.0
    case.close()


@pytest.mark.parametrize("bcname", ["walls", "slip", "periodic", "inout", "outin-y"])
def test_vorticity_2d(gpu, oracle, bcname):
    from tests.test_dim2_gpu import BC2, Case2
    from varden_amd import advance as adv
    K = Case2((24, 20), BC2[bcname], seed=12)
    L = oracle.lib()
    u, _ = K.random_state()
    ov = K.ofab(0, 2)
    L.vo_makevort(ov.ref, 1, u.ref, K.odx, C.byref(K.obc))
    L.vo_makemagvel(ov.ref, 0, u.ref)
    gu, gv = K.gmf(u), K.gmf(K.ofab(0, 2))
    adv.make_magvel(gv, 0, gu)
    adv.make_vorticity(gv, 1, gu, K.dx, K.bct)
    assert_bits(gv.to_numpy(), ov.a, "2-D magvel / vort " + bcname)
    K.close()
    gpu.initialize(params_for(WALLS), 0, 1, 0)                 # back to dm = 3 for the tests that follow


def _state(G):
    out = []
    for name in ("uold", "sold", "gp", "p"):
        for mf in getattr(G, name):
            g, nd = mf.ng, mf.nodal
            for li in range(mf.nfabs()):
                a = mf.to_numpy(li)
                out.append(a[g:a.shape[0] - g, g:a.shape[1] - g, g:a.shape[2] - g].copy())
    return out


def test_plotfile_holds_the_device_data(gpu, oracle, tmp_path):
    from varden_amd import driver, plotfile
    G = driver.VardenAMR(16, [((8, 8, 8), (23, 23, 15)), ((8, 8, 16), (23, 23, 23))], WALLS, params=params_for(WALLS, cflfac=0.9))
    G.step(); G.step()
    name = plotfile.write_plotfile(G, base=str(tmp_path / "plt"))
    assert name.endswith("plt00002")
    r = plotfile.read_ml_multifab(name)
    assert r["names"] == ["x_vel", "y_vel", "z_vel", "density", "tracer", "magvel", "vort", "gpx", "gpy", "gpz"]
    assert r["nlevs"] == 2 and r["time"] == G.time and r["rr"] == [2] and r["pd"] == ((0, 0, 0), (15, 15, 15))
    assert r["levels"][1]["boxes"] == [((8, 8, 8), (23, 23, 15)), ((8, 8, 16), (23, 23, 23))]
    for n in range(2):
        for li, a in enumerate(r["levels"][n]["fabs"]):
            u = G.uold[n].to_numpy(li)[3:-3, 3:-3, 3:-3]
            s = G.sold[n].to_numpy(li)[3:-3, 3:-3, 3:-3]
            gp = G.gp[n].to_numpy(li)[1:-1, 1:-1, 1:-1]
            assert_bits(a[..., 0:3], u, "plotfile velocity"); assert_bits(a[..., 3:5], s, "plotfile scalars"); assert_bits(a[..., 7:10], gp, "plotfile gp")
            assert_bits(a[..., 5], np.sqrt(u[..., 0] * u[..., 0] + u[..., 1] * u[..., 1] + u[..., 2] * u[..., 2]), "plotfile magvel")
            assert np.isfinite(a[..., 6]).all() and a[..., 6].min() >= 0.0 and a[..., 6].max() > 0.0
    G.close()


@pytest.mark.parametrize("decomp", [(1, 1, 1), (2, 2, 1)])
def test_restart_continues_bit_for_bit(gpu, oracle, tmp_path, decomp):
    from varden_amd import driver, plotfile
    prm = lambda: params_for(WALLS, cflfac=0.9, visc_coef=0.001)   # noqa: E731
    A = driver.Varden(32, WALLS, prm(), init_shrink=0.1, init_iter=1, decomp=decomp)
    A.step(); A.step()
    chk = plotfile.write_checkfile(A, base=str(tmp_path / "chk"))
    A.step(); A.step()
    ref = _state(A)
    tA, dtA = A.time, A.dt
    A.close()
    c = plotfile.read_checkfile(chk)
    assert c["nlevs"] == 1 and len(c["boxes"][0]) == decomp[0] * decomp[1] * decomp[2]
    B = driver.Varden(32, WALLS, prm(), decomp=decomp, restart=c, restart_step=2)
    assert B.istep == 2 and B.time == c["time"] and B.dt == c["dt"]
    B.step(); B.step()
    assert B.istep == 4 and B.time == tA and B.dt == dtA
    for x, y in zip(ref, _state(B)):
        assert_bits(y, x, "state after restart, decomp %r" % (decomp,))
    B.close()


def test_restart_of_a_hierarchy(gpu, oracle, tmp_path):
    from varden_amd import driver, plotfile
    fine = [((8, 8, 8), (23, 23, 15)), ((8, 8, 16), (23, 23, 23))]
    A = driver.VardenAMR(16, fine, WALLS, params=params_for(WALLS, cflfac=0.9), init_iter=1, do_initial_projection=1)
    A.step()
    chk = plotfile.write_checkfile(A, base=str(tmp_path / "chk"))
    A.step(); A.step()
    ref = _state(A)
    A.close()
    c = plotfile.read_checkfile(chk)
    assert c["nlevs"] == 2 and c["rr"] == [2] and c["boxes"][1] == fine
    B = driver.VardenAMR(16, c["boxes"][1], WALLS, params=params_for(WALLS, cflfac=0.9), base_boxes=c["boxes"][0], restart=c, restart_step=1)
    B.step(); B.step()
    for x, y in zip(ref, _state(B)):
        assert_bits(y, x, "hierarchy state after restart")
    B.close()


def test_fixed_dt_and_stop_time(gpu, oracle):
    """dt control of src/varden.f90:196-199, 318-326: fixed_dt overrides estdt; the last step is cut so that the run ends at stop_time"""
    from varden_amd import driver
    A = driver.Varden(32, WALLS, params_for(WALLS, cflfac=0.9), init_shrink=0.1, init_iter=1)
    dt0 = A.dt
    A.close()
    T = 2.5 * dt0
    B = driver.Varden(32, WALLS, params_for(WALLS, cflfac=0.9), init_shrink=0.1, init_iter=1, fixed_dt=dt0, stop_time=T)
    dts = []
    while B.time < T:
        B.step()
        dts.append(B.dt)
    assert len(dts) == 3 and dts[0] == dt0 and dts[1] == dt0 and abs(dts[2] - 0.5 * dt0) < 1e-12 * dt0 and B.time == T
    B.close()
