"""What one ghost exchange costs on ONE MI355X, by message size and path (profiles/r03_exchange_budget.md is built from this).

    python tools/exchange_probe.py            runs itself three times (child processes):
      local    box-to-box copies of one rank                       (k_xcopy: what a single-rank run does)
      packed1  VDN_FORCE_PACKED=1: pack -> device memcpy -> unpack  (the packed path without a transport)
      packed2  VDN_FORCE_PACKED=2: pack -> ncclSend/ncclRecv on a 1-rank RCCL communicator (the rank's own buffer) -> unpack

Geometry: two boxes of n^3 side by side in x, walls; a cell-centred multifab (1 component, 1 ghost layer: the multigrid halo of a
colour pass), a nodal one (the halo of a Jacobi sweep) and the 3-component ng = 3 state (uold: the widest fill of a step).  Each
exchange of the pair moves two face messages -- what ONE rank of a 2 x 1 x 1 decomposition sends and receives -- through one ncclGroup.
Times are host wall clock over back-to-back calls (launch overhead included, as in the N > 1 path, which issues every exchange from
the host), and the device time of the same sequence between two events.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(mode):
    from varden_amd import boxlib as bl, capi
    from varden_amd.capi import default_params
    prm = default_params()
    bl.initialize(prm, 0, 1, 0)
    if mode == "packed2":
        bl.comm_init(bl.comm_get_unique_id())
    lib = capi.load()
    out = []
    for n in (256, 128, 64, 32, 16, 8):
        pd = ((0, 0, 0), (2 * n - 1, n - 1, n - 1))
        boxes = [((0, 0, 0), (n - 1, n - 1, n - 1)), ((n, 0, 0), (2 * n - 1, n - 1, n - 1))]
        mla = bl.MLLayout([pd], [boxes])
        for name, nc, ng, nodal in (("cell nc1 ng1", 1, 1, (0, 0, 0)), ("nodal nc1 ng1", 1, 1, (1, 1, 1)), ("cell nc3 ng3", 3, 3, (0, 0, 0))):
            mf = bl.MultiFab(mla, 0, nc, ng, nodal)
            mf.setval(1.0, all=True)
            for _ in range(5):
                mf.fill_boundary()
            lib.vdn_device_synchronize()
            reps = 200
            st = (C.c_long * 24)()
            lib.vdn_comm_stats(st, 1)
            t0 = time.perf_counter()
            for _ in range(reps):
                mf.fill_boundary()
            t_issue = time.perf_counter() - t0
            lib.vdn_device_synchronize()
            t_all = time.perf_counter() - t0
            lib.vdn_comm_stats(st, 1)
            face = (n + nodal[1]) * (n + nodal[2]) * ng * nc * 8 * (2 if nodal[0] else 1)   # bytes one box sends (a nodal face plane is shared: 2 planes)
            out.append({"mode": mode, "n": n, "field": name, "bytes_per_message": face, "us_per_exchange": round(1e6 * t_all / reps, 2),
                        "us_host_issue": round(1e6 * t_issue / reps, 2), "exchanges_counted": int(st[0]), "sends_counted": int(st[1]),
                        "doubles_sent": int(st[2])})
            mf.destroy()
        mla.destroy()
    print("EXCHANGE_PROBE " + json.dumps(out))
    bl.comm_finalize()
    bl.finalize()


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    rows = []
    for mode, env in (("local", {}), ("packed1", {"VDN_FORCE_PACKED": "1"}), ("packed2", {"VDN_FORCE_PACKED": "2"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, VDN_LIB_FLAVOUR="testing", **env), capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print("mode %s failed:\n%s\n%s" % (mode, p.stdout[-2000:], p.stderr[-2000:]))
            continue
        for line in p.stdout.splitlines():
            if line.startswith("EXCHANGE_PROBE "):
                rows += json.loads(line[len("EXCHANGE_PROBE "):])
    print("%-8s %-14s %5s %12s %12s %12s" % ("mode", "field", "n", "bytes/msg", "us/exchange", "us host"))
    for r in rows:
        print("%-8s %-14s %5d %12d %12.2f %12.2f" % (r["mode"], r["field"], r["n"], r["bytes_per_message"], r["us_per_exchange"], r["us_host_issue"]))
    json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "exchange_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
