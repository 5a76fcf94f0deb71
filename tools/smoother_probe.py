"""rocprofv3 target: the MAC-MG smoother colour pass at 256^3 (the roofline kernel), plus a copy kernel of
known byte count to calibrate FETCH_SIZE / WRITE_SIZE for 8-byte-per-lane accesses (MI355X_MICROARCH.md, HBM)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, boxlib as bl, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bl.initialize(capi.default_params(), 0, 1, 0)
lo, hi = (0, 0, 0), (n - 1,) * 3
mla = bl.MLLayout([(lo, hi)], [[(lo, hi)]])
rh, phi = bl.MultiFab(mla, 0, 1, 0), bl.MultiFab(mla, 0, 1, 1)
beta = [bl.MultiFab(mla, 0, 1, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
rng = np.random.default_rng(0)
r = rng.standard_normal((n, n, n, 1)); rh.from_numpy(r - r.mean())
rho = bl.MultiFab(mla, 0, 1, 1)
ra = rng.uniform(1.0, 10.0, size=rho.shape(0))
rho.from_numpy(ra)
for d in range(3):      # beta = 2 / (rho_i + rho_i-1) on the faces, as mk_mac_coeffs builds it
    hi_sl, lo_sl = [slice(1, -1)] * 3, [slice(1, -1)] * 3
    hi_sl[d], lo_sl[d] = slice(1, None), slice(0, -1)
    beta[d].from_numpy(2.0 / (ra[tuple(hi_sl)] + ra[tuple(lo_sl)]))
stored = len(sys.argv) > 3 and sys.argv[3] == "stored"
ms, cells = adv.bench_cc_smoother(rh, phi, beta, [1.0 / n] * 3, [[bl.BC_NEU] * 2] * 3, nl, rho=None if stored else rho)
print("smoother (%s): %.5f ms/launch, %d cells, %.1f GB/s algorithmic (48 B/cell)" % ("stored beta" if stored else "beta from rho", ms, cells, 48.0 * cells / ms / 1e6))
# calibration: k_copy of one component of an ng=0 multifab = n^3*8 B read + n^3*8 B written, 8 B/lane
a, b = bl.MultiFab(mla, 0, 1, 0), bl.MultiFab(mla, 0, 1, 0)
a.setval(1.0)
for _ in range(5):
    b.copy_c(0, a, 0, 1, 0)
print("calibration k_copy: %d bytes read, %d bytes written per launch" % (n ** 3 * 8, n ** 3 * 8))
