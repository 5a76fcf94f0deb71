"""times the slopes launch of velpred (three components) and of the scalar advance (two) at 256^3 through the unit-test hook vdn_k_slope ... not available per launch form;
instead: steps the bubble and prints the kernel's mean from the library's own phase timers is too coarse -- use rocprofv3 --kernel-trace --stats on this script."""
import sys
sys.path.insert(0, ".")
from varden_amd import driver, capi
from varden_amd.capi import default_params
G = driver.Varden(256, [[15, 15]] * 3, default_params(cflfac=0.9), init_shrink=0.1, init_iter=1, swap_state=True)
for _ in range(6):
    G.step()
capi.load().vdn_device_synchronize()
