"""two viscous (and diffusive) steps on one box against the oracle, with a hash of the new state (tests/test_advance_gpu.py::test_viscous_solves_by_colour): how the finest level of
the three velocity solves (and the tracer's) is stored comes from VDN_MAC_SPLIT / VDN_MAC_SPLIT_MIN / VDN_MAC_SLAB (read once per process).  argv: bc-set name, nx ny nz, diff_coef"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import voracle as vo
    from tests.util import BC_SETS, params_for
    from varden_amd import advance as adv, capi, driver
    sets = dict(BC_SETS, slipz=[[15, 15], [14, 14], [14, 15]], periodicx=[[-1, -1], [15, 15], [15, 15]])
    phys = sets[sys.argv[1]]
    n = tuple(int(v) for v in sys.argv[2:5])
    diff = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    kw = dict(cflfac=0.9, visc_coef=0.001, diff_coef=diff)
    h = 1.0 / max(n)
    hi = tuple(n[d] * h for d in range(3))
    # the oracle's own initial data (vo_initdata places the bubble in index space of the given spacing) handed to the library, as smoke() does
    ou, os_ = vo.Fab((0, 0, 0), tuple(v - 1 for v in n), 3, 3), vo.Fab((0, 0, 0), tuple(v - 1 for v in n), 3, 2)
    vo.lib().vo_initdata(ou.ref, os_.ref, vo.dvec([h] * 3), 1)
    G = driver.Varden(n, phys, params_for(phys, **kw), prob_type=1, grav=-9.8, prob_hi=hi, init_shrink=0.1, init_iter=1, u0=ou.a, s0=os_.a)
    form0 = capi.load().vdn_last_mac_level_form()          # after the start-up's pressure iteration: the form of its last cell-centred solve (a viscous one)
    hsh = hashlib.sha256()
    if os.environ.get("VDN_WORKER_ORACLE", "1") != "0":
        O = vo.Sim(n, phys, params_for(phys, **kw), prob_type=1, grav=-9.8, prob_hi=hi, init_shrink=0.1, init_iter=1)
    else:
        O = None
    for _ in range(2):
        G.step()
        if O is not None:
            O.step()
            assert G.dt == O.dt, (G.dt, O.dt)
            assert (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]) == (int(O.mgstat[0].cycles), int(O.mgstat[1].cycles))
            for gm, om, g in ((G.unew[0], O.unew, 3), (G.snew[0], O.snew, 3)):
                a, b = gm.to_numpy()[g:-g, g:-g, g:-g], om.valid()
                assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-300), np.abs(a - b).max()
    for m in (G.unew[0], G.snew[0], G.p[0]):
        hsh.update(np.ascontiguousarray(m.to_numpy()).tobytes())
    print("FORM", form0, capi.load().vdn_last_mac_level_form())
    print("HASH", hsh.hexdigest(), G.dt)
    G.close()


if __name__ == "__main__":
    main()
