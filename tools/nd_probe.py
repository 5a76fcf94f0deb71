"""the nodal Jacobi sweep at n^3 in isolation (VDN_ND_BENCH hook of nd_solve; VDN_ND_DBG: 1 no arithmetic, 2 no loads in the march)"""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("VDN_ND_BENCH", "40")
os.environ["VDN_NO_GRAPHS"] = "1"
from varden_amd import driver
from varden_amd.capi import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = driver.Varden(n, [[15, 15]] * 3, default_params(cflfac=0.9), init_shrink=0.1, init_iter=0, do_initial_projection=1)
G.close()
