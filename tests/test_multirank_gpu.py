"""SURVEY.md §8(e) on ONE GPU: 2 and 4 ranks (processes) run the driver's step loop on a decomposed domain and must reproduce,
bit for bit, the single-rank run on the same boxes -- ghost exchange through per-peer packed buffers, multigrid halos on every
level, agglomerated coarse levels (all-gather), residual norms and estdt maxima (all-reduce MAX).  RCCL refuses two ranks on
one device, so the transport is the test double tests/fake_rccl/libfake_rccl.so (named by VDN_RCCL_LIB); the real RCCL entry
points are exercised on hardware by tests/test_multibox_gpu.py::test_rccl_self_exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def rank_groups(nranks, per_proc):
    """the ranks each worker process hosts: a GPU box admits six processes on its card, so eight ranks run as four processes of two rank
    threads (tests/_rank_threads.py: one private copy of the library per rank)"""
    return [",".join(str(r) for r in range(a, min(a + per_proc, nranks))) for a in range(0, nranks, per_proc)]


def run_ranks(tmp_path, tag, nranks, decomp, n, nsteps, periodic, real_rccl=False, per_proc=1):
    if nranks > 1 and not os.path.exists(FAKE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(FAKE)])
    idfile, prefix = str(tmp_path / (tag + ".id")), str(tmp_path / tag)
    if real_rccl:                                # one GPU per rank, the RCCL torch ships (dlopen of librccl.so.1)
        env = dict(os.environ, VDN_WORKER_DEVICE_PER_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("VDN_RCCL_LIB", None)
    else:
        # VDN_OVERLAP=1: halo exchange on the second stream + shell kernels on every level (by default only boxes of >= 2^20 cells do)
        env = dict(os.environ, VDN_RCCL_LIB=FAKE, VDN_TESTING="1", FAKE_RCCL_DIR=str(tmp_path), VDN_OVERLAP=os.environ.get("VDN_OVERLAP", "1"))
        if nranks >= 8:
            env["FAKE_RCCL_MAXMSG_MB"] = "8"     # 64 mailboxes: keep the memory-mapped file small (the messages of 32^3 boxes are a few hundred KB)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_multirank_worker.py"), g, str(nranks), idfile, prefix]
                              + [str(x) for x in decomp] + [str(x) for x in n] + [str(nsteps), str(int(periodic))], env=env, cwd=ROOT)
             for g in rank_groups(nranks, per_proc)]
    try:
        rcs = [p.wait(timeout=400) for p in procs]
    finally:
        for p in procs:                      # exact PIDs of the children this test started
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * len(procs), rcs
    out = {}
    for r in range(nranks):
        with np.load(prefix + ".%d.npz" % r) as z:
            for k in z.files:
                if k == "dt":
                    out.setdefault("dt", z[k])
                    assert np.array_equal(out["dt"], z[k]), "ranks disagree on dt"
                else:
                    out[k] = z[k]
    return out


@pytest.mark.parametrize("nranks,decomp,n,periodic", [(2, (2, 1, 1), (64, 32, 32), False), (2, (2, 1, 1), (64, 32, 32), True),
                                                      (4, (2, 2, 1), (64, 64, 32), False)])
def test_ranks_reproduce_single_rank_bits(gpu, tmp_path, nranks, decomp, n, periodic):
    ref = run_ranks(tmp_path, "ref", 1, decomp, n, 2, periodic)
    got = run_ranks(tmp_path, "mr", nranks, decomp, n, 2, periodic)
    assert sorted(ref) == sorted(got)
    assert np.array_equal(ref["dt"], got["dt"]), (ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())
    assert np.isfinite(got["u0"]).all() and np.abs(got["u0"]).max() > 0


@pytest.mark.parametrize("periodic", [False, True])
def test_eight_ranks_2x2x2_reproduce_single_rank_bits(gpu, tmp_path, periodic):
    """BASELINE.json configs[2]'s rank topology (2 x 2 x 2, one 32^3 box per rank; velpred.f90:102-119, macproject.f90:117,492, estdt.f90:69 are the
    exchange sites): every rank has SEVEN peers -- three across faces, three across edges and one across the corner, the last four only in
    the nodal halos and the state's ghost shells --, the agglomerated multigrid tails are gathered from eight ranks, three multigrid levels per
    solver stay distributed.  Start-up sequence + two steps, bit for bit against one rank on the same eight boxes; periodic: the x direction
    wraps, so the two ranks of a row are each other's neighbour on both sides.  Eight ranks = four processes of two rank threads."""
    ref = run_ranks(tmp_path, "ref", 1, (2, 2, 2), (64, 64, 64), 2, periodic)
    got = run_ranks(tmp_path, "mr8", 8, (2, 2, 2), (64, 64, 64), 2, periodic, per_proc=2)
    assert sorted(ref) == sorted(got) and len([k for k in got if k.startswith("u")]) == 8
    assert np.array_equal(ref["dt"], got["dt"]), (ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())
    assert np.isfinite(got["u0"]).all() and np.abs(got["u7"]).max() > 0


def test_two_rank_threads_in_one_process_equal_two_processes(gpu, tmp_path):
    """the harness itself: two ranks as threads of ONE process (private library copies) give the bits of two ranks in two processes"""
    a = run_ranks(tmp_path, "pp", 2, (2, 1, 1), (64, 32, 32), 2, False)
    b = run_ranks(tmp_path, "tt", 2, (2, 1, 1), (64, 32, 32), 2, False, per_proc=2)
    assert sorted(a) == sorted(b)
    for k in sorted(a):
        assert np.array_equal(a[k], b[k]), k


def test_two_ranks_of_128_cubed_with_the_default_overlap_rule(gpu, tmp_path, monkeypatch):
    """boxes of 2^21 cells: large enough for the default rule (VDN_OVERLAP unset) to put the halo traffic of the finest multigrid level on the
    second stream and finish its face cells in the shell kernels, for the paired density pass of the MAC solve (n >= 128) and for the fused
    Godunov marches with interior box faces; the coarser levels take the serial path.  One step, against the single-rank bits."""
    monkeypatch.setenv("VDN_OVERLAP", "-1")                 # run_ranks passes it on: -1 = the library's own rule
    ref = run_ranks(tmp_path, "ref", 1, (2, 1, 1), (256, 128, 128), 1, False)
    got = run_ranks(tmp_path, "mr", 2, (2, 1, 1), (256, 128, 128), 1, False)
    assert np.array_equal(ref["dt"], got["dt"]), (ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())


def test_two_ranks_with_the_finest_mac_level_by_colour(gpu, tmp_path, monkeypatch):
    """round 6: macproject's finest level stored by colour on two ranks (VDN_MAC_SPLIT_MIN=0: from any size; by default from 2^23 cells per rank, i.e. configs[2]'s
    256^3 box per GPU): the ghost entries of one colour travel through the packed buffers of the split arrays' own plan, on the halo stream next to the interior
    cells, the shell kernel behind them (the library's own overlap rule), the coarse correction inside the first sweep, residual + restriction per box.
    One step of two 128^3 boxes against ONE rank with the level interleaved (round 5's form): the same bits."""
    monkeypatch.setenv("VDN_OVERLAP", "-1")
    monkeypatch.setenv("VDN_MAC_SPLIT", "0")
    ref = run_ranks(tmp_path, "ref", 1, (2, 1, 1), (256, 128, 128), 1, False)
    monkeypatch.delenv("VDN_MAC_SPLIT")
    monkeypatch.setenv("VDN_MAC_SPLIT_MIN", "0")
    got = run_ranks(tmp_path, "mr", 2, (2, 1, 1), (256, 128, 128), 1, False)
    assert np.array_equal(ref["dt"], got["dt"]), (ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())
    forms = sorted(open(str(tmp_path / ("mr.%d.form" % r))).read().strip() for r in range(2))
    assert forms == ["1", "1"], forms          # (vdn_last_mac_level_form on both ranks: the split level ran)


def _ngpus():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU


@pytest.mark.skipif(_ngpus() < 2, reason="needs two GPUs: real RCCL refuses two ranks on one device")
@pytest.mark.parametrize("periodic", [False, True])
def test_two_gpus_real_rccl_reproduce_single_rank_bits(gpu, tmp_path, periodic):
    """2 ranks on 2 REAL GPUs under the real RCCL transport (ncclSend/ncclRecv groups, ncclAllReduce(MAX),
    ncclAllGather of the agglomerated multigrid tail) reproduce the single-rank run on the same two boxes bit for bit"""
    ref = run_ranks(tmp_path, "ref", 1, (2, 1, 1), (64, 32, 32), 2, periodic)
    got = run_ranks(tmp_path, "rccl", 2, (2, 1, 1), (64, 32, 32), 2, periodic, real_rccl=True)
    assert sorted(ref) == sorted(got) and np.array_equal(ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())


def _bench_line(args, env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    import json
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"], ["--config", "amr2"]])
def test_bench_spawns_its_own_ranks(gpu, tmp_path, extra):
    """`python bench.py --gpus 2` from a bare shell (no torchrun, no WORLD_SIZE) starts two ranks itself and prints ONE line with
    n_gpus = 2 and rccl_nranks = 2 read back from the communicator.  On a one-GPU box both ranks share the device
    (VDN_BENCH_ONE_DEVICE) and the transport is the RCCL test double; with two GPUs present the real RCCL runs"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if _ngpus() < 2:
        if not os.path.exists(FAKE):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(FAKE)])
        env.update(VDN_BENCH_ONE_DEVICE="1", VDN_RCCL_LIB=FAKE, VDN_TESTING="1", FAKE_RCCL_DIR=str(tmp_path))
    line = _bench_line(["--gpus", "2", "--steps", "2", "--warmup", "1", "--box", "32", "--skip-cpu"] + extra, env)
    assert line["n_gpus"] == 2 and line["rccl_nranks"] == 2 and line["value"] > 0
    assert line["scaling"] == ("weak" if not extra else "strong")
    one = _bench_line(["--gpus", "1", "--steps", "2", "--warmup", "1", "--box", "32", "--skip-cpu"] + extra, env)
    assert one["n_gpus"] == 1 and one["rccl_nranks"] == 1
    if extra:                                   # strong scaling / fixed hierarchy: the same global problem on 1 and 2 ranks
        assert one["config"]["cells"] == line["config"]["cells"]
        if "amr2" not in extra:                 # (the two-rank hierarchy cuts its base level into 8 boxes: per-box eps of the Godunov dead-band)
            assert one["config"]["vcycles_per_step"] == line["config"]["vcycles_per_step"]
    else:
        assert 2 * one["config"]["cells"] == line["config"]["cells"]


def run_amr_ranks(tmp_path, tag, nranks, nlev, visc, mode="fixed", extra=(), per_proc=1):
    if nranks > 1 and not os.path.exists(FAKE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(FAKE)])
    idfile, prefix = str(tmp_path / (tag + ".id")), str(tmp_path / tag)
    env = dict(os.environ, VDN_RCCL_LIB=FAKE, VDN_TESTING="1", FAKE_RCCL_DIR=str(tmp_path), VDN_OVERLAP=os.environ.get("VDN_OVERLAP", "1"))
    if nranks >= 8:
        env["FAKE_RCCL_MAXMSG_MB"] = "8"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_multirank_amr_worker.py"), g, str(nranks), idfile, prefix, str(nlev), str(visc), mode] + list(extra),
                              env=env, cwd=ROOT) for g in rank_groups(nranks, per_proc)]
    try:
        rcs = [p.wait(timeout=400) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * len(procs), rcs
    out = {}
    for r in range(nranks):
        with np.load(prefix + ".%d.npz" % r) as z:
            for k in z.files:
                if k in ("dt", "nboxes", "nregrids"):
                    out.setdefault(k, z[k])
                    assert np.array_equal(out[k], z[k]), "ranks disagree on " + k
                else:
                    out[k] = z[k]
    return out


def test_tagged_grids_and_regrid_on_two_ranks(gpu, tmp_path):
    """inputs_bubble_3d's flow on two ranks: tag_boxes + make_new_grids with the tag bitmap all-reduced, boxes dealt by cell count,
    regridding every second step (fillpatch, nodal prolongation and the old -> new copies through views) -- same boxes and same bits as
    one rank"""
    ref = run_amr_ranks(tmp_path, "tref", 1, 2, 0.001, "tagged")
    got = run_amr_ranks(tmp_path, "tmr", 2, 2, 0.001, "tagged")
    assert ref["nregrids"][0] >= 1 and np.array_equal(ref["nboxes"], got["nboxes"]) and np.array_equal(ref["dt"], got["dt"])
    assert sorted(ref) == sorted(got)
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())
    # the plot and checkpoint files of the two ranks (one Cell_D file per rank) read back as the files of the single rank
    from varden_amd import plotfile
    for kind, sub in (("_plt00004", ""), ("_chk00004", "State"), ("_chk00004", "Pressure")):
        a = plotfile.read_ml_multifab(os.path.join(str(tmp_path / ("tref" + kind)), sub))
        b = plotfile.read_ml_multifab(os.path.join(str(tmp_path / ("tmr" + kind)), sub))
        assert a["nlevs"] == b["nlevs"] and a["time"] == b["time"] and a["names"] == b["names"]
        for La, Lb in zip(a["levels"], b["levels"]):
            assert La["boxes"] == Lb["boxes"]
            for x, y in zip(La["fabs"], Lb["fabs"]):
                assert np.array_equal(x, y)
    assert os.path.exists(str(tmp_path / "tmr_plt00004" / "Level_01" / "Cell_D_00001"))
    assert open(str(tmp_path / "tref_chk00004" / "Header")).read() == open(str(tmp_path / "tmr_chk00004" / "Header")).read()
    # restart on two ranks from the checkpoint the two ranks wrote == restart on one rank from the one-rank checkpoint (steps 5, 6; regrid at 5)
    r1 = run_amr_ranks(tmp_path, "rref", 1, 2, 0.001, "restart", [str(tmp_path / "tref_chk00004")])
    r2 = run_amr_ranks(tmp_path, "rmr", 2, 2, 0.001, "restart", [str(tmp_path / "tmr_chk00004")])
    assert sorted(r1) == sorted(r2) and np.array_equal(r1["dt"], r2["dt"]) and np.array_equal(r1["nboxes"], r2["nboxes"])
    for k in sorted(r1):
        assert np.array_equal(r1[k], r2[k]), "restart: %s differs" % k


def test_tagged_three_level_hierarchy_on_eight_ranks(gpu, tmp_path):
    """BASELINE.json configs[4]'s shape on eight ranks (four processes of two rank threads): a tagged three-level hierarchy (tag_boxes.f90:65-94,
    32^3 base in four boxes, max_grid_size 16), its boxes dealt to the ranks by cell count -- so some ranks own nothing on level 0 --, viscous,
    start-up sequence, four steps with a regrid every second one: same boxes, same dt, same bits as one rank"""
    ref = run_amr_ranks(tmp_path, "t3ref", 1, 3, 0.001, "tagged")
    got = run_amr_ranks(tmp_path, "t3mr", 8, 3, 0.001, "tagged", per_proc=2)
    assert len(ref["nboxes"]) == 3 and ref["nregrids"][0] >= 1
    assert np.array_equal(ref["nboxes"], got["nboxes"]) and np.array_equal(ref["dt"], got["dt"]) and sorted(ref) == sorted(got)
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())


@pytest.mark.parametrize("nranks,nlev,visc", [(2, 2, 0.0), (3, 2, 0.001), (2, 3, 0.001), (5, 3, 0.001)])
def test_amr_ranks_reproduce_single_rank(gpu, tmp_path, nranks, nlev, visc):
    """SURVEY.md section 8(e), "multi-level extras", on ONE GPU: the boxes of every level of a fixed hierarchy are dealt to 2 / 3 ranks by cell
    count (a fine box and the coarse boxes under it usually sit on different ranks); start-up sequence + two steps (composite MAC,
    viscous and nodal solves, average-down, coarse-fine ghost interpolation, flux matching -- all through windows of remote boxes)
    reproduce the single-rank run on the same boxes bit for bit.  Five ranks: more ranks than boxes on every level (4, 3, 2 boxes), so
    some ranks own nothing on a level and still take part in every collective.  Transport: the RCCL test double (see the module docstring)."""
    ref = run_amr_ranks(tmp_path, "aref", 1, nlev, visc)
    got = run_amr_ranks(tmp_path, "amr", nranks, nlev, visc)
    assert sorted(ref) == sorted(got)
    assert np.array_equal(ref["dt"], got["dt"]), (ref["dt"], got["dt"])
    for k in sorted(ref):
        assert np.array_equal(ref[k], got[k]), "%s differs: max %.3e" % (k, np.abs(ref[k] - got[k]).max())
    assert np.isfinite(got["u0_0"]).all() and np.abs(got["u1_0"]).max() > 0
