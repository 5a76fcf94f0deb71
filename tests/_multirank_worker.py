"""one or several ranks of a multi-rank run on ONE GPU (tests/test_multirank_gpu.py): ranks are processes -- or threads of a process, each on
its private copy of the library (tests/_rank_threads.py: an eight-rank run fits the six-process limit of a GPU box as 4 x 2) --, the
transport is the RCCL test double tests/fake_rccl (VDN_RCCL_LIB), the rendezvous a file.
argv: rank[,rank...] nranks idfile outprefix bx by bz nx ny nz nsteps periodic"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ranks, nranks = [int(r) for r in sys.argv[1].split(",")], int(sys.argv[2])
    from tests._rank_threads import run_ranks
    run_ranks(ranks, lambda rank, pkg: one_rank(rank, nranks, pkg), os.path.dirname(sys.argv[4]))


def one_rank(rank, nranks, pkg):
    idfile, outprefix = sys.argv[3], sys.argv[4]
    decomp = tuple(int(x) for x in sys.argv[5:8])
    n = tuple(int(x) for x in sys.argv[8:11])
    nsteps, periodic = int(sys.argv[11]), int(sys.argv[12])
    bl, driver = pkg.boxlib, pkg.driver
    prm = pkg.capi.default_params(cflfac=0.9)
    dev = rank if os.environ.get("VDN_WORKER_DEVICE_PER_RANK") == "1" else 0     # real RCCL: one GPU per rank
    comm_id = None
    if nranks > 1:
        bl.initialize(prm, rank, nranks, dev)
        if rank == 0:
            cid = bl.comm_get_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(cid)
            os.rename(idfile + ".tmp", idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            time.sleep(0.01)
            assert time.time() - t0 < 120, "rendezvous timed out"
        comm_id = open(idfile, "rb").read()
    walls = [[bl.NO_SLIP_WALL] * 2] * 3
    if periodic:
        walls = [[bl.PERIODIC] * 2] + [[bl.NO_SLIP_WALL] * 2] * 2
    h = 1.0 / max(n)
    G = driver.Varden(n, walls, prm, prob_type=1, grav=-9.8, prob_hi=tuple(n[d] * h for d in range(3)), init_shrink=0.1, init_iter=1,
                      device=dev, decomp=decomp, rank=rank, nranks=nranks, comm_id=comm_id)
    dts = []
    for _ in range(nsteps):
        G.step()
        dts.append(G.dt)
    out = {"dt": np.array(dts)}
    for li, gi in enumerate(G.local):
        out["u%d" % gi] = G.unew[0].to_numpy(li)[3:-3, 3:-3, 3:-3]
        out["s%d" % gi] = G.snew[0].to_numpy(li)[3:-3, 3:-3, 3:-3]
        out["p%d" % gi] = G.p[0].to_numpy(li)[1:-1, 1:-1, 1:-1]
    np.savez(outprefix + ".%d.npz" % rank, **out)
    with open(outprefix + ".%d.form" % rank, "w") as f:        # how the last MAC solve kept its finest level (vdn_last_mac_level_form)
        f.write("%d\n" % pkg.capi.load().vdn_last_mac_level_form())
    G.close()


if __name__ == "__main__":
    main()
