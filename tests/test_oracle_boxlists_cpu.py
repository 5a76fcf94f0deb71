"""The oracle's hierarchies of BOX LISTS (oracle/vo.h, round 5: level arrays with a cell mask, face fields and the Godunov kernels box by box) checked on
the CPU against the one-box form of rounds 2-4 and against size-independent properties.  The GPU comparison on the tagged grids of BASELINE.json
configs[3] / [4] is tests/test_amr_gpu.py::test_tagged_hierarchy_against_the_box_list_oracle."""
import numpy as np
import pytest

from oracle import voracle as vo
from varden_amd.capi import default_params

WALLS = [[15, 15]] * 3


def run(boxes, nc=16, steps=2, **kw):
    S = vo.SimML(nc, boxes, WALLS, prm=default_params(cflfac=0.9, **kw), init_shrink=0.1, init_iter=1, do_initial_projection=1)
    cyc = []
    m0 = composite_mass(S)
    for _ in range(steps):
        S.step(); cyc.append((S.mgstat[0].cycles, S.mgstat[1].cycles))
    return S, cyc, m0, composite_mass(S)


def composite_mass(S):
    m = 0.0
    for n in range(S.nlev):
        msk = S.levels[n].mask()
        if n + 1 < S.nlev:
            f = S.levels[n + 1]
            fm = f.mask()[::2, ::2, ::2]
            o = [f.lo[d] // 2 - S.levels[n].lo[d] for d in range(3)]
            cov = np.zeros_like(msk)
            cov[o[0]:o[0] + fm.shape[0], o[1]:o[1] + fm.shape[1], o[2]:o[2] + fm.shape[2]] = fm
            msk = msk & ~cov
        m += S.sold[n].valid()[..., 0][msk].sum() / 8.0 ** n
    return m


ONE = ((8, 8, 8), (23, 23, 23))
TWO = [((8, 8, 8), (15, 23, 23)), ((16, 8, 8), (23, 23, 23))]
EIGHT = [((8 + 8 * i, 8 + 8 * j, 8 + 8 * k), (15 + 8 * i, 15 + 8 * j, 15 + 8 * k)) for k in range(2) for j in range(2) for i in range(2)]
LSHAPE = [((8, 8, 8), (23, 15, 23)), ((8, 16, 8), (15, 23, 23))]


@pytest.mark.parametrize("name,split,kw", [("two", TWO, {}), ("eight", EIGHT, {}), ("eight-viscous", EIGHT, dict(visc_coef=0.01, diff_coef=0.005))])
def test_a_box_cut_into_boxes_runs_like_the_one_box(name, split, kw):
    """the same rectangle as one box and as a list of boxes: the level arrays, masks, per-direction interface values, per-box Godunov, face fields per
    box and their exchanges must reproduce the one-box run -- to 1e-10 (the per-box dead band of the upwinding, velpred.f90:1965-1980, is the one
    thing that may differ), with the FAC iteration counts of both composite solves equal"""
    S1, c1, _, _ = run([ONE], **kw)
    S2, c2, _, _ = run([split], **kw)
    assert c1 == c2 and S1.dt == S2.dt, (c1, c2, S1.dt, S2.dt)
    for n in range(2):
        for nm in ("uold", "sold", "p", "gp"):
            a, b = getattr(S1, nm)[n].valid(), getattr(S2, nm)[n].valid()
            assert np.abs(a - b).max() <= 1e-10 * max(np.abs(a).max(), 1e-300), "%s: %s level %d differs by %.3e" % (name, nm, n, np.abs(a - b).max())


def test_unions_that_are_not_rectangles_and_composite_mass():
    """an L-shaped refined level (a re-entrant interface edge: one ghost position reached from two directions with two interpolated values) and a third
    level of two boxes on it: the composite solves converge, everything stays finite, the symmetric bubble stays symmetric across the plane the union is
    symmetric about (z), and the mass of the composite grid -- uncovered cells of every level -- is conserved to round-off because the coarse cells next
    to a finer level take the fine fluxes (mkflux.f90:137-146)"""
    fine2 = [((20, 20, 20), (35, 27, 43)), ((20, 28, 20), (27, 43, 43))]
    for boxes in ([LSHAPE], [LSHAPE, fine2]):
        S, cyc, m0, m1 = run(boxes)
        assert all(c[0] < 30 and c[1] < 40 for c in cyc), cyc
        assert abs(m1 - m0) <= 1e-13 * m0, (m1 - m0) / m0
        for n in range(S.nlev):
            msk = S.levels[n].mask()
            for f in (S.uold[n], S.sold[n]):
                assert np.isfinite(f.valid()[msk]).all()
        r = S.sold[1].valid()[..., 0]                       # the union and the bubble are symmetric under x <-> y (gravity acts along z)
        msk = S.levels[1].mask()
        rt, mt = np.transpose(r, (1, 0, 2)), np.transpose(msk, (1, 0, 2))
        if S.nlev == 2:                                     # (the third level's boxes below are not symmetric)
            assert np.array_equal(msk, mt) and np.abs((r - rt)[msk]).max() <= 1e-8


def test_flux_restriction_is_what_conserves_the_composite_mass():
    """a bubble that sits ACROSS the coarse-fine interface (the fine level ends inside it): the density fluxes through the interface are those of the
    bubble itself, and the composite mass still does not drift"""
    S = vo.SimML(16, [((8, 8, 8), (19, 23, 23))], WALLS, prm=default_params(cflfac=0.9), prob_type=1, init_shrink=0.1, init_iter=1, do_initial_projection=1)
    m0 = composite_mass(S)
    for _ in range(4):
        S.step()
    m1 = composite_mass(S)
    assert abs(m1 - m0) <= 1e-13 * m0, (m1 - m0) / m0
    # the bubble (radius 0.1 around the centre, cell 16 of 32 on the fine level) is cut by the fine level's high-x face at fine cell 19
    assert S.sold[1].valid()[-1, :, :, 0].max() > 1.5


def test_regridding_onto_the_same_and_onto_other_box_lists():
    """SimML.regrid (build_and_fill_data of src/regrid.f90:269-339 on level arrays): onto the SAME box lists the state comes back bit for bit (every cell and
    node is copied from the old level); onto a smaller level the remaining cells keep their data and the run goes on; onto a larger one the new cells are
    the fillpatch interpolation of the level below -- a linear density stays linear there"""
    S, _, _, _ = run([TWO], steps=1)
    before = [f.valid().copy() for f in S.uold + S.sold + S.gp + S.p]
    S.regrid([TWO])
    after = [f.valid() for f in S.uold + S.sold + S.gp + S.p]
    msk = [S.levels[n].mask() for n in range(2)] * 3 + [S.node_mask(n) for n in range(2)]
    for a, b, m in zip(before, after, msk):
        assert np.array_equal(a[m], b[m])
    S.regrid([[TWO[0]]])                                  # the level shrinks to its first box
    assert S.levels[1].hi == TWO[0][1]
    S.step()
    assert np.isfinite(S.uold[1].valid()).all() and S.mgstat[0].cycles < 30 and S.mgstat[1].cycles < 40
    # growth: a field that is linear in x on both levels is reproduced on the cells the level gains
    for n in range(2):
        h = 1.0 / (16 << n)
        x = (np.arange(S.levels[n].lo[0] - 3, S.levels[n].hi[0] + 4) + 0.5) * h
        S.sold[n].a[..., 0] = 1.0 + x[:, None, None]
    S.regrid([EIGHT])
    h = 1.0 / 32
    x = (np.arange(S.levels[1].lo[0], S.levels[1].hi[0] + 1) + 0.5) * h
    assert np.abs(S.sold[1].valid()[..., 0] - (1.0 + x[:, None, None])).max() <= 1e-14
