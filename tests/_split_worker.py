"""macproject on one 132 x 36 x 40 box (tests/test_projection_gpu.py::test_split_colour_level_*): argv = bc-set name [nx ny nz].  Compares with the oracle and prints a
hash of the projected MAC velocities; how the finest level of the solve is stored comes from VDN_MAC_SPLIT / VDN_MAC_SPLIT_MIN / VDN_MAC_KFLIP (read once per
process).  132 cells: 33 lane pairs of the split pass, a clamped tail of the wave."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import voracle as oracle
    from tests.util import BC_SETS, Case
    from tests.test_projection_gpu import face_fabs
    from varden_amd import advance as adv
    sets = dict(BC_SETS, zout=[[15, 12], [14, 15], [11, 12]],           # outlets at x-hi and z-hi, inlet at z-lo
                periodicx=[[-1, -1], [15, 15], [15, 15]], periodicyz=[[15, 12], [-1, -1], [-1, -1]])
    n = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (132, 36, 40)
    nb = tuple(int(v) for v in sys.argv[5:8]) if len(sys.argv) >= 8 else (1, 1, 1)      # round 6: the domain cut into nb boxes (tests/test_multibox_gpu.py: Split)
    case = Case(n, sets[sys.argv[1]], seed=13, iso=True)
    if nb != (1, 1, 1):
        return multibox(case, nb)
    L = oracle.lib()
    _, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    L.vo_fill_boundary(s.ref, case.opm)
    L.vo_physbc(s.ref, 0, 3, case.prm.nscal, C.byref(case.obc), C.byref(case.prm))
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
    assert cyc == st.cycles, (cyc, st.cycles)
    h = hashlib.sha256()
    for d in range(3):
        g, o = gum[d].to_numpy(), oum[d].a
        scale = np.abs(o[1:-1, 1:-1, 1:-1]).max()
        err = np.abs(g - o)[1:-1, 1:-1, 1:-1].max()
        assert err <= 1e-11 * scale, (d, err, scale)
        h.update(np.ascontiguousarray(g[1:-1, 1:-1, 1:-1]).tobytes())
    from varden_amd import capi
    print("FORM", capi.load().vdn_last_mac_level_form())
    print("HASH", h.hexdigest(), cyc)
    case.close()


def multibox(case, nb):
    """the same projection on the domain cut into boxes: the oracle (one box) once -- VDN_WORKER_ORACLE=0 skips it --, the hash over the gathered faces"""
    from oracle import voracle as oracle
    from tests.test_multibox_gpu import Split
    from tests.test_projection_gpu import face_fabs
    from varden_amd import advance as adv
    from varden_amd import capi
    L = oracle.lib()
    _, s = case.random_state()
    s.a[..., 0] = np.abs(s.a[..., 0]) + 0.5
    L.vo_fill_boundary(s.ref, case.opm)
    L.vo_physbc(s.ref, 0, 3, case.prm.nscal, C.byref(case.obc), C.byref(case.prm))
    u, _ = case.random_state()
    force = case.ofab(1, 3)
    oum = face_fabs(case, 1, 1, 1.0e20)
    L.vo_velpred(u.ref, oracle.fab_ptr_array(oum), force.ref, case.odx, C.c_double(0.2 * min(case.dx)), C.byref(case.obc), C.byref(case.prm))
    for f in oum:
        L.vo_fill_boundary(f.ref, case.opm)
    mac_rhs = case.ofab(1, 1)
    sp = Split(case, nb)
    gum = [sp.scatter(f) for f in oum]
    adv.macproject(sp.mla, [gum], [sp.scatter(s)], [sp.scatter(mac_rhs)], [case.dx], sp.bct, case.obc.press_comp + 1)
    cyc, r0, r = adv.last_solver_stats("mac")
    got = [sp.gather(gum[d], oum[d]) for d in range(3)]
    if os.environ.get("VDN_WORKER_ORACLE", "1") != "0":
        st = oracle.CMgStat()
        L.vo_macproject(oracle.fab_ptr_array(oum), s.ref, mac_rhs.ref, case.odx, C.byref(case.obc), case.opm, C.byref(case.prm), C.byref(st))
        assert cyc == st.cycles, (cyc, st.cycles)
        for d in range(3):
            o = oum[d].a
            scale = np.abs(o[1:-1, 1:-1, 1:-1]).max()
            err = np.abs(got[d] - o)[1:-1, 1:-1, 1:-1].max()
            assert err <= 1e-11 * scale, (d, err, scale)
    h = hashlib.sha256()
    for d in range(3):
        assert np.isfinite(got[d][1:-1, 1:-1, 1:-1]).all()
        h.update(np.ascontiguousarray(got[d][1:-1, 1:-1, 1:-1]).tobytes())
    print("FORM", capi.load().vdn_last_mac_level_form())
    print("HASH", h.hexdigest(), cyc)
    sp.close()
    case.close()


if __name__ == "__main__":
    main()
