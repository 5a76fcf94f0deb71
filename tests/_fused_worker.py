"""prints a hash of phi after 3 red-black sweeps on a 72 x 64 x 40 box (tests/test_kernels_gpu.py::test_fused_sweeps_equal_colour_passes);
the smoother variant is chosen by VDN_FUSED_GSRB in the environment (read once per process)"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varden_amd import advance as adv, boxlib as bl, capi  # noqa: E402

n = (72, 64, 40)
bl.initialize(capi.default_params(), 0, 1, 0)
lo, hi = (0, 0, 0), tuple(x - 1 for x in n)
mla = bl.MLLayout([(lo, hi)], [[(lo, hi)]])
rh, phi = bl.MultiFab(mla, 0, 1, 0), bl.MultiFab(mla, 0, 1, 1)
beta = [bl.MultiFab(mla, 0, 1, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
rng = np.random.default_rng(5)
r = rng.standard_normal(n + (1,))
rh.from_numpy(r - r.mean())
for d in range(3):
    beta[d].from_numpy(rng.uniform(0.1, 1.0, size=beta[d].shape(0)))
bc = [[bl.BC_NEU, bl.BC_DIR], [bl.BC_NEU, bl.BC_NEU], [bl.BC_DIR, bl.BC_NEU]]
adv.cc_smooth(rh, phi, beta, [1.0 / 64] * 3, bc, 3)
a = phi.to_numpy()[1:-1, 1:-1, 1:-1]
assert np.isfinite(a).all() and np.abs(a).max() > 0
print("HASH", hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
