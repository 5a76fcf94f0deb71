"""A/B of one library switch: `python tools/ab_switch.py VDN_X 0 1 [n] [steps]` runs the n^3 bubble for `steps` steps with the switch at each value (child
processes: switches are read once), prints ms per step, the phase split and a hash of the final state -- equal hashes = the same bits."""
import hashlib
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, vals = sys.argv[1], sys.argv[2:4]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
code = textwrap.dedent("""
    import sys, time, hashlib
    sys.path.insert(0, %r)
    import numpy as np
    from varden_amd import driver, advance as adv, capi
    from varden_amd.capi import default_params
    G = driver.Varden(%d, [[15, 15]] * 3, default_params(cflfac=0.9), init_shrink=0.1, init_iter=1, swap_state=True)
    for _ in range(2): G.step()
    capi.load().vdn_device_synchronize()
    ph = dict(scalar=0.0, velocity=0.0, mac=0.0, hg=0.0, total=0.0)
    t0 = time.perf_counter()
    for _ in range(%d):
        G.step()
        for k, v in adv.last_step_timing().items(): ph[k] += v
    capi.load().vdn_device_synchronize()
    el = time.perf_counter() - t0
    h = hashlib.sha256()
    for m in (G.uold[0], G.sold[0], G.p[0], G.gp[0]):
        h.update(np.ascontiguousarray(m.to_numpy(0)).tobytes())
    print("RESULT ms/step %%.3f  phases %%s  hash %%s" %% (1e3 * el / %d, {k: round(1e3 * v / %d, 3) for k, v in ph.items()}, h.hexdigest()[:16]))
""" % (ROOT, n, steps, steps, steps))
for v in vals:
    env = dict(os.environ, VDN_LIB_FLAVOUR="testing"); env[name] = v      # (the switches live in the testing build)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    print(name, "=", v, ":", [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")] or r.stderr[-1500:], flush=True)
