"""how fast are the nodal sweeps on a level that fits the Infinity Cache?  steps a 256 x 256 x NZ bubble (argv[1], default 64); run under
rocprofv3 --kernel-trace --stats and compare kk_nd_march_pair's time per plane with the 0.514 us per plane of the 257^3 level (132 us)."""
import sys
sys.path.insert(0, ".")
from varden_amd import driver, capi
from varden_amd.capi import default_params
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = driver.Varden((256, 256, nz), [[15, 15]] * 3, default_params(cflfac=0.9), prob_hi=(1.0, 1.0, nz / 256.0), init_shrink=0.1, init_iter=1, swap_state=True)
for _ in range(4):
    G.step()
capi.load().vdn_device_synchronize()
