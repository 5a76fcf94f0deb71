"""plot / checkpoint file layout on the host (no GPU): what write_ml_multifab puts on disk is what read_ml_multifab returns, for
cell-centred and nodal data on several levels and boxes; header fields follow the BoxLib plotfile layout; the checkpoint header
is the &chkpoint namelist of src/checkpoint.f90:36-38,74-78."""
import os

import numpy as np

from varden_amd import plotfile as pf


def _levels(rng, nodal=(0, 0, 0), nc=3):
    boxes = [[((0, 0, 0), (7, 7, 7))], [((4, 4, 4), (11, 9, 11)), ((4, 10, 4), (11, 11, 7))]]
    out = []
    for lb in boxes:
        fabs = [np.asfortranarray(rng.standard_normal(tuple(hi[d] - lo[d] + 1 + nodal[d] for d in range(3)) + (nc,))) for lo, hi in lb]
        out.append(dict(boxes=lb, nodal=nodal, fabs=fabs))
    return out


def test_ml_multifab_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    for nodal in ((0, 0, 0), (1, 1, 1)):
        lv = _levels(rng, nodal)
        d = str(tmp_path / ("mf%d" % nodal[0]))
        pf.write_ml_multifab(d, lv, [2], 3, names=["a", "b", "c"], pd=((0, 0, 0), (7, 7, 7)), prob_lo=[0, 0, 0], prob_hi=[1, 1, 1],
                             time=0.125, dx=[0.125] * 3)
        r = pf.read_ml_multifab(d)
        assert r["names"] == ["a", "b", "c"] and r["dm"] == 3 and r["nlevs"] == 2 and r["time"] == 0.125 and r["rr"] == [2]
        assert r["pd"] == ((0, 0, 0), (7, 7, 7)) and r["dx"] == [0.125] * 3
        for L, R in zip(lv, r["levels"]):
            assert R["boxes"] == L["boxes"] and tuple(R["nodal"]) == nodal
            for a, b in zip(L["fabs"], R["fabs"]):
                assert np.array_equal(a, b)


def test_header_layout(tmp_path):
    rng = np.random.default_rng(4)
    d = str(tmp_path / "plt00007")
    pf.write_ml_multifab(d, _levels(rng, nc=2), [2], 3, names=["x_vel", "density"], pd=((0, 0, 0), (7, 7, 7)), prob_lo=[0, 0, 0],
                         prob_hi=[1, 1, 1], time=1.5, dx=[0.125] * 3)
    h = open(os.path.join(d, "Header")).read().split("\n")
    assert h[0] == "NavierStokes-V1.1" and h[1] == "2" and h[2:4] == ["x_vel", "density"] and h[4] == "3"
    assert float(h[5]) == 1.5 and h[6] == "1"                        # time, finest level
    assert h[9] == "2"                                               # refinement ratios
    assert h[10] == "((0,0,0) (7,7,7) (0,0,0)) ((0,0,0) (15,15,15) (0,0,0))"
    assert [float(x) for x in h[12].split()] == [0.125] * 3 and [float(x) for x in h[13].split()] == [0.0625] * 3
    assert "Level_00/Cell" in h and "Level_01/Cell" in h
    ch = open(os.path.join(d, "Level_01", "Cell_H")).read().split("\n")
    assert ch[:4] == ["1", "0", "2", "0"] and ch[4] == "(2 0" and ch[5] == "((4,4,4) (11,9,11) (0,0,0))"
    assert ch[9].startswith("FabOnDisk: Cell_D_00000 0")
    raw = open(os.path.join(d, "Level_01", "Cell_D_00000"), "rb").read(200)
    assert raw.startswith(b"FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))((4,4,4) (11,9,11) (0,0,0)) 2\n")


def test_two_dimensional_boxes(tmp_path):
    rng = np.random.default_rng(5)
    lv = [dict(boxes=[((0, 0, 0), (15, 7, 0))], nodal=(1, 1, 0), fabs=[np.asfortranarray(rng.standard_normal((17, 9, 1, 1)))])]
    d = str(tmp_path / "p2")
    pf.write_ml_multifab(d, lv, [], 2)
    r = pf.read_ml_multifab(d)
    assert r["dm"] == 2 and r["levels"][0]["boxes"] == lv[0]["boxes"] and tuple(r["levels"][0]["nodal"]) == (1, 1, 0)
    assert np.array_equal(r["levels"][0]["fabs"][0], lv[0]["fabs"][0])
    assert "((0,0) (16,8) (1,1))" in open(os.path.join(d, "Level_00", "Cell_H")).read()


def test_two_writers_one_level_set(tmp_path):
    """the several-ranks layout, two writers one after the other: each rank's fabs in its own Cell_D file, offsets from the box sizes,
    minima / maxima combined by the MAX reduction, text files by rank 0 -- the reader sees one hierarchy"""
    rng = np.random.default_rng(6)
    lv = _levels(rng, nc=2)
    owner = [[0], [1, 0]]
    d = str(tmp_path / "two")
    seen, turn = [], [0]

    def keep(a):                           # rank 1 goes first: remember what it contributes, level by level
        seen.append(a.copy())
        return a

    def combine(a):                        # rank 0: the MAX reduction with rank 1's contribution of the same level
        turn[0] += 1
        return np.maximum(a, seen[turn[0] - 1])

    for rank, red in ((1, keep), (0, combine)):
        mine = [dict(boxes=L["boxes"], nodal=L["nodal"], owner=ow, fabs={g: L["fabs"][g] for g in range(len(ow)) if ow[g] == rank}) for L, ow in zip(lv, owner)]
        pf.write_ml_multifab(d, mine, [2], 3, pd=((0, 0, 0), (7, 7, 7)), nc=2, rank=rank, reduce_max=red)
    r = pf.read_ml_multifab(d)
    for L, R in zip(lv, r["levels"]):
        assert R["boxes"] == L["boxes"]
        for a, b in zip(L["fabs"], R["fabs"]):
            assert np.array_equal(a, b)
    ch = open(os.path.join(d, "Level_01", "Cell_H")).read().split("\n")
    assert ch[9] == "FabOnDisk: Cell_D_00001 0" and ch[10] == "FabOnDisk: Cell_D_00000 0"
    mins = [float(x) for x in ch[13].split(",")[:2]]
    assert mins == [lv[1]["fabs"][0][..., c].min() for c in range(2)]


class _FakeSim:
    """the attributes write_grids / write_job_info read from a driver object"""
    dm, nc, rank, nranks, phys = 3, 16, 0, 1, [[15, 15], [14, 14], [-1, -1]]
    boxes = [[((0, 0, 0), (15, 15, 15))], [((8, 8, 8), (23, 23, 15)), ((8, 8, 16), (23, 23, 23))]]
    local = [[0], [0, 1]]


def test_grids_file_round_trip_and_job_info(tmp_path):
    """write_grids (src/varden.f90:621-662) appends a block per call; read_grids (read_a_hgproj_grid) takes the first one"""
    g = str(tmp_path / "grids.out")
    pf.write_grids(g, _FakeSim, 0)
    pf.write_grids(g, _FakeSim, 4)
    text = open(g).read()
    assert text.count("At step") == 2 and "   ((0, 0, 0) (15, 15, 15) (0,0,0))    1\n" in text and "   ((0, 0, 0) (31, 31, 31) (0,0,0))    2\n" in text
    domains, levels = pf.read_grids(g)
    assert domains == [((0, 0, 0), (15, 15, 15)), ((0, 0, 0), (31, 31, 31))] and levels == _FakeSim.boxes
    pf.write_job_info(str(tmp_path), _FakeSim, inputs_text="&PROBIN\n max_levs = 2\n/", job_name="bubble")
    info = open(str(tmp_path / "job_info")).read()
    assert "job name:    bubble" in info and "level: 2" in info and "number of boxes = 2" in info
    assert "-x: no slip wall" in info and "+y: slip wall" in info and "-z: periodic" in info and "max_levs = 2" in info
