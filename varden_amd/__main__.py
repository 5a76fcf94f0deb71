"""python -m varden_amd <inputs file> [--steps N] [--outdir DIR] [--device D]: the reference executable's command line (src/main.f90
reads the inputs file named by the first argument and calls varden()).  Plot and checkpoint files go under --outdir."""
import argparse
import os
import time

from . import advance as adv
from . import inputs


def main():
    ap = argparse.ArgumentParser(prog="python -m varden_amd")
    ap.add_argument("inputs_file")
    ap.add_argument("--steps", type=int, default=None, help="override max_step")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)

    def report(G):                                           # the step line of src/varden.f90:347,  1000 format
        print("STEP = %d  TIME = %.15g  DT = %.15g   (mac %d, hg %d cycles)" % (G.istep, G.time, G.dt, adv.last_solver_stats("mac")[0],
                                                                               adv.last_solver_stats("hg")[0]), flush=True)

    t0 = time.time()
    nl, G = inputs.run(open(a.inputs_file).read(), a.steps, report, device=a.device, outdir=a.outdir)
    print("Total Run time (s) = %.3f" % (time.time() - t0))
    for f in G.files_written:
        print("wrote", f)
    G.close()


if __name__ == "__main__":
    main()
