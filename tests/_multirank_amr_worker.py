"""one or several ranks (threads, tests/_rank_threads.py) of a multi-rank AMR run on ONE GPU (tests/test_multirank_gpu.py::test_amr_ranks_reproduce_single_rank): a two- or
three-level fixed hierarchy whose boxes are dealt to the ranks by cell count.  argv: rank[,rank...] nranks idfile outprefix nlev visc [tagged | restart <checkpoint>]"""
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
    nlev, visc = int(sys.argv[5]), float(sys.argv[6])
    tagged = len(sys.argv) > 7 and sys.argv[7] in ("tagged", "restart")
    restart = len(sys.argv) > 7 and sys.argv[7] == "restart"          # continue from the checkpoint <argv[8]> written by a "tagged" run
    bl, driver, plotfile = pkg.boxlib, pkg.driver, pkg.plotfile
    prm = pkg.capi.default_params(cflfac=0.9, visc_coef=visc)
    comm_id = None
    if nranks > 1:
        bl.initialize(prm, rank, nranks, 0)
        if rank == 0:
            with open(idfile + ".tmp", "wb") as f:
                f.write(bl.comm_get_unique_id())
            os.rename(idfile + ".tmp", idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            time.sleep(0.01)
            assert time.time() - t0 < 120, "rendezvous timed out"
        comm_id = open(idfile, "rb").read()
    walls = [[bl.NO_SLIP_WALL] * 2] * 3
    base = [((0, 0, 0), (7, 7, 15)), ((8, 0, 0), (15, 7, 15)), ((0, 8, 0), (7, 15, 15)), ((8, 8, 0), (15, 15, 15))]    # level 0 in four equal boxes (the single-level multigrid wants equal boxes)
    fine = [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 15, 23)), ((16, 16, 8), (23, 23, 23))]          # partial shared faces
    finer = [[((24, 24, 24), (31, 39, 39)), ((32, 24, 24), (39, 39, 39))]] if nlev == 3 else []
    if restart:          # grids and state from the checkpoint, every rank reads the boxes it owns (initialize_from_restart)
        chk = plotfile.read_checkfile(sys.argv[8])
        G = driver.VardenAMR(32, chk["boxes"][1], walls, params=prm, finer_levels=chk["boxes"][2:], base_boxes=chk["boxes"][0], regrid_int=2, max_levs=nlev,
                             max_grid_size=16, rank=rank, nranks=nranks, comm_id=comm_id, restart=chk, restart_step=4)
    elif tagged:         # grids from the tagged bubble on a 32^3 base in four boxes, regridding every second step (inputs_bubble_3d)
        base = [((0, 0, 0), (15, 15, 31)), ((16, 0, 0), (31, 15, 31)), ((0, 16, 0), (15, 31, 31)), ((16, 16, 0), (31, 31, 31))]
        levels = driver.VardenAMR.tagged_grids(32, walls, prm, max_levs=nlev, buf_wid=2, max_grid_size=16, rank=rank, nranks=nranks, comm_id=comm_id, base_boxes=base)
        G = driver.VardenAMR(32, levels[0], walls, params=prm, finer_levels=levels[1:], base_boxes=base, init_iter=1, do_initial_projection=1,
                             regrid_int=2, max_levs=nlev, max_grid_size=16, rank=rank, nranks=nranks, comm_id=comm_id)
    else:
        G = driver.VardenAMR(16, fine, walls, params=prm, finer_levels=finer, base_boxes=base, init_iter=1, do_initial_projection=1,
                             rank=rank, nranks=nranks, comm_id=comm_id)
    dts = []
    for _ in range(2 if restart else (4 if tagged else 2)):
        G.step()
        dts.append(G.dt)
    out = {"dt": np.array(dts), "nboxes": np.array([len(b) for b in G.boxes]), "nregrids": np.array([G.nregrids])}
    for n in range(G.nlev):
        for li, gi in enumerate(G.local[n]):
            out["u%d_%d" % (n, gi)] = G.unew[n].to_numpy(li)[3:-3, 3:-3, 3:-3]
            out["s%d_%d" % (n, gi)] = G.snew[n].to_numpy(li)[3:-3, 3:-3, 3:-3]
            out["p%d_%d" % (n, gi)] = G.p[n].to_numpy(li)[1:-1, 1:-1, 1:-1]
    if tagged and not restart:                             # plot and checkpoint files, every rank its own Cell_D file
        plotfile.write_plotfile(G, base=outprefix + "_plt")
        plotfile.write_checkfile(G, base=outprefix + "_chk")
    np.savez(outprefix + ".%d.npz" % rank, **out)
    G.close()


if __name__ == "__main__":
    main()
