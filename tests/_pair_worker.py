"""one viscous advance_timestep on a 256 x 128 x 128 bubble (tests/test_kernels_gpu.py::test_paired_colour_pass_*): argv = bx by bz outfile.
The colour pass of the MAC multigrid's finest level is chosen by VDN_GSRB_PAIR (read once per process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    decomp, out = tuple(int(x) for x in sys.argv[1:4]), sys.argv[4]
    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    from varden_amd import driver
    from varden_amd.capi import default_params
    walls = [[bl.NO_SLIP_WALL] * 2] * 3
    G = driver.Varden((256, 128, 128), walls, default_params(cflfac=0.9, visc_coef=0.001), prob_hi=(2.0, 1.0, 1.0), init_shrink=0.1, init_iter=0,
                      do_initial_projection=0, decomp=decomp)
    G.step()
    np.savez(out, u=G.gather_valid(G.unew[0]), s=G.gather_valid(G.snew[0]), cyc=np.array(adv.last_solver_stats("mac")[0]))
    G.close()


if __name__ == "__main__":
    main()
