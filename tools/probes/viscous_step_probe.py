"""how much of a VISCOUS 256^3 step (visc_coef = 0.001, as every 3-D input of exec/test has it) goes into the three Crank-Nicolson velocity solves?
usage (under rocprofv3 --kernel-trace --stats for the kernel split): python tools/probes/viscous_step_probe.py [n=256] [nsteps=5]"""
import sys, time
sys.path.insert(0, ".")
import torch
from varden_amd import advance as adv, driver
from varden_amd.capi import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W = [[15, 15]] * 3
for visc in (0.0, 0.001):
    G = driver.Varden(n, W, default_params(cflfac=0.9, visc_coef=visc), prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1, swap_state=True)
    for _ in range(2):
        G.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ph = dict(scalar=0.0, velocity=0.0, mac=0.0, hg=0.0, total=0.0)
    for _ in range(ns):
        G.step()
        for k, v in adv.last_step_timing().items():
            ph[k] += v
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("visc_coef %g: %.2f ms per step; phases %s" % (visc, 1e3 * el / ns, {k: round(1e3 * v / ns, 2) for k, v in ph.items()}), flush=True)
    G.close()
