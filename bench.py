#!/usr/bin/env python
"""bench.py -- cells*steps/sec of advance_timestep on the MI355X-native hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--box 256] [--skip-cpu]

Workload (BASELINE.json configs[1]): 3-D 256^3 single-level variable-density bubble
(reference src/initdata.f90:212-238, exec/test/inputs_bubble_3d with visc_coef = 0), one 256^3 box,
no-slip walls, gravity -9.8, cflfac 0.9, init_shrink 0.1, init_iter 1; MAC + HG projection every
step.  A "step" = one pass of the reference's time-loop body (src/varden.f90:291-328): ghost fills,
estdt, advance_timestep, uold<-unew.  All state is resident in HBM before the timed region.

One JSON line is printed by rank 0.  `roofline` is the MAC-multigrid red-black Gauss-Seidel colour
pass on the finest level (48 algorithmic B/cell/pass), timed with HIP events on the launch
stream inside the library; `cpu_baseline` is the CPU oracle (a port, OpenMP) on a bounded sample.
For N > 1 the domain is decomposed: one 256^3 box per rank/GPU (weak scaling; N = 8 is the 512^3 / 2x2x2
case of BASELINE.json configs[2]), ghost cells and multigrid halos exchanged with RCCL point-to-point
over xGMI, norms / estdt by ncclAllReduce, coarse multigrid levels agglomerated (all-gather).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--box", dest="n", type=int, default=256)
    ap.add_argument("--cpu-box", dest="cpu_n", type=int, default=128)
    ap.add_argument("--skip-cpu", dest="no_cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_dev = os.environ.get("VDN_BENCH_ONE_DEVICE") == "1"     # debugging aid: every rank on GPU 0, gloo control plane
    if one_dev:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"

    from varden_amd import advance as adv
    from varden_amd import boxlib as bl
    from varden_amd import driver
    from varden_amd.capi import default_params

    n = args.n
    walls = [[bl.NO_SLIP_WALL] * 2] * 3
    prm = default_params(cflfac=0.9)
    decomp = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world)
    assert decomp is not None, "bench.py supports 1, 2, 4 or 8 GPUs"
    nglob = tuple(n * decomp[d] for d in range(3))
    h = 1.0 / (n * max(decomp))                            # dx = dy = dz; N = 8 gives the unit cube at 512^3
    prob_hi = tuple(nglob[d] * h for d in range(3))
    comm_id = None
    if world > 1:                                          # RCCL unique id: rank 0 creates it, everybody receives it
        bl.initialize(prm, rank, world, local_rank)
        idt = torch.zeros(128, dtype=torch.uint8, device="cpu" if one_dev else "cuda")
        if rank == 0:
            idt.copy_(torch.tensor(list(bl.comm_get_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        comm_id = bytes(idt.cpu().tolist())
    G = driver.Varden(nglob, walls, prm, prob_type=1, grav=-9.8, prob_hi=prob_hi, init_shrink=0.1, init_iter=1,
                      device=local_rank, decomp=decomp, rank=rank, nranks=world, comm_id=comm_id)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        G.step()
    barrier()
    t0 = time.perf_counter()
    phases = dict(scalar=0.0, velocity=0.0, mac=0.0, hg=0.0, total=0.0)
    cyc = dict(mac=0, hg=0)
    for _ in range(args.steps):
        G.step()
        for k, v in adv.last_step_timing().items():
            phases[k] += v
        cyc["mac"] += adv.last_solver_stats("mac")[0]
        cyc["hg"] += adv.last_solver_stats("hg")[0]
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cpu" if one_dev else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    cells = n ** 3 * world
    value = cells * args.steps / el
    rho = G.sold[0].to_numpy()[..., 0] if rank == 0 else None      # rank 0's box, for the smoother probe's coefficients
    G.close()                                                     # also tears the RCCL communicator down

    # ---- roofline of the dominant kernel: one colour pass of the MAC-MG smoother at n^3 ----------
    roof = None
    if rank == 0:
        bl.initialize(prm, 0, 1, local_rank)               # the probe is a single-rank, single-box measurement
        lo0, hi0 = (0, 0, 0), (n - 1,) * 3
        mla = bl.MLLayout([(lo0, hi0)], [[(lo0, hi0)]])
        rh, phi = bl.MultiFab(mla, 0, 1, 0), bl.MultiFab(mla, 0, 1, 1)
        beta = [bl.MultiFab(mla, 0, 1, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
        import numpy as np
        for d in range(3):      # beta = 2/(rho_i + rho_{i-1})  (macproject.f90:376-394), built on the host for the probe
            sl_hi = [slice(3, -3)] * 3
            sl_lo = [slice(3, -3)] * 3
            sl_hi[d] = slice(3, rho.shape[d] - 2)
            sl_lo[d] = slice(2, rho.shape[d] - 3)
            beta[d].from_numpy((2.0 / (rho[tuple(sl_hi)] + rho[tuple(sl_lo)]))[..., None])
        rng = np.random.default_rng(0)
        r = rng.standard_normal((n, n, n, 1))
        rh.from_numpy(r - r.mean())
        bc = [[bl.BC_NEU] * 2] * 3
        rho_mf = bl.MultiFab(mla, 0, 1, 3)
        rho_mf.from_numpy(rho[..., None])
        # the pass macproject runs on its finest level (face coefficients recomputed from rho), and the stored-coefficient pass next to it
        ms, ncell = adv.bench_cc_smoother(rh, phi, beta, [1.0 / n] * 3, bc, 200, rho=rho_mf)
        ms_stored, _ = adv.bench_cc_smoother(rh, phi, beta, [1.0 / n] * 3, bc, 200)
        rho_mf.destroy()
        alg_bytes = 48.0 * ncell
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md), collected
        # separately with `rocprofv3 --pmc` on tools/smoother_probe.py and committed under profiles/
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_smoother_rho_pmc.json")        # the kernel timed above (kk_cc_gsrb_rho_pair)
        if n == 256 and os.path.exists(pmc):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
        roof = {"bound": "hbm", "kernel": "kk_cc_gsrb_rho_pair (MAC-MG red-black GS colour pass, %d^3; beta recomputed from rho, 2x2 cells per thread: ~34 B/cell of "
                                          "real traffic against the 48 B/cell algorithmic figure; stored-beta pass kk_cc_gsrb: %.5f ms)" % (n, ms_stored),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(ms, 5), "alg_bytes_per_launch": alg_bytes}
        for m in [rh, phi] + beta:
            m.destroy()

    # ---- CPU baseline: the oracle (a port of the same algorithm, OpenMP) on a bounded sample ------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:      # reported at N = 1 only
        nthreads = min(16, os.cpu_count() or 1)
        os.environ["OMP_NUM_THREADS"] = str(nthreads)
        from oracle import voracle as vo
        cn = args.cpu_n
        O = vo.Sim(cn, walls, default_params(cflfac=0.9), prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1)
        tc = time.perf_counter()
        O.step()
        tcpu = time.perf_counter() - tc
        cpu = {"value": round(cn ** 3 / tcpu, 1), "unit": "cells*steps/s", "cores": nthreads, "kind": "port",
               "sample": "%d^3 bubble (same problem, smaller box), 1 timed step after the initial pressure iteration; "
                         "gcc -O2 -fopenmp, OMP_NUM_THREADS=%d" % (cn, nthreads)}

    if rank == 0:
        out = {
            "metric": "cells*steps/sec on advance_timestep",
            "value": round(value, 1), "unit": "cells*steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D %d^3 single-level variable-density bubble, 1 box/GPU, MAC+HG projection each step "
                                   "(BASELINE.json configs[1])" % n,
                       "parallelism": "single GPU" if world == 1 else
                                      "domain decomposition %dx%dx%d, one %d^3 box per GPU (global %dx%dx%d), RCCL p2p ghost exchange + allreduce" % (decomp + (n,) + nglob),
                       "phase_ms_per_step": {k: round(1e3 * v / args.steps, 3) for k, v in phases.items()},
                       "vcycles_per_step": {k: round(v / args.steps, 2) for k, v in cyc.items()}},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
