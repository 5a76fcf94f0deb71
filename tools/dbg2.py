import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from tests.util import BC_SETS, Case
from oracle import voracle as oracle
from varden_amd import advance as adv
bcname = sys.argv[1] if len(sys.argv) > 1 else "periodic"
case = Case((16, 12, 8), BC_SETS[bcname], seed=2, use_minion=0)
u, _ = case.random_state()
force = case.ofab(1, 3)
force.a[...] = case.rng.standard_normal(force.a.shape)
dt = 0.4 * min(case.dx)
def face_fabs(case, ng, nc, val=0.0):
    return [case.ofab(ng, nc, tuple(1 if t == d else 0 for t in range(3)), val) for d in range(3)]
oum = face_fabs(case, 1, 1, 1.0e20)
oracle.lib().vo_velpred(u.ref, oracle.fab_ptr_array(oum), force.ref, case.odx, C.c_double(dt), C.byref(case.obc), C.byref(case.prm))
pre = [f.a.copy() for f in oum]
for f in oum:
    oracle.lib().vo_fill_boundary(f.ref, case.opm)
gum = [case.gmf(f) for f in face_fabs(case, 1, 1, 1.0e20)]
adv.velpred(case.gmf(u), gum, case.gmf(force), case.dx, dt, case.bct)
for d in range(3):
    g, o = gum[d].to_numpy(), oum[d].a
    bad = np.argwhere(g != o)
    print("dir", d, "mismatches", len(bad))
    for b in bad[:12]:
        print("   idx", tuple(int(x) - 1 for x in b[:3]), "gpu", g[tuple(b)], "oracle", o[tuple(b)])
    # periodic consistency of the valid faces in the oracle
    sl_lo = [slice(1, -1)] * 3; sl_hi = [slice(1, -1)] * 3
    sl_lo[d] = 1; sl_hi[d] = -2
    print("   oracle lo-face == hi-face:", np.array_equal(pre[d][tuple(sl_lo)], pre[d][tuple(sl_hi)]))
