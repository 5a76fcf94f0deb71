"""Plot and checkpoint files of the driver (src/varden.f90:492-620 write_plotfile / write_checkfile, src/checkpoint.f90,
src/restart.f90, src/initialize.f90:22-88 initialize_from_restart), in the BoxLib multi-level multifab layout that FBoxLib's
fabio_ml_multifab_write_d produces:

    <dir>/Header                      text: variables, domain, levels, box list per level
    <dir>/Level_NN/Cell_H             text: boxes of the level, file + byte offset of every fab, per-fab minima / maxima
    <dir>/Level_NN/Cell_D_00000       binary: for each fab a one-line ASCII "FAB ..." header, then its doubles (x fastest, then y, z, comp)

fabio itself is not part of the reference tree (EXT): the layout above is written from the published BoxLib format and is
[EXT-UNVERIFIED] byte for byte (directory names, number formats); what the tests pin is that `read_ml_multifab` returns exactly what
`write_ml_multifab` was given and that a run restarted from a checkpoint continues bit for bit.

Several ranks: every rank writes the fabs it owns into its own Cell_D_<rank:05d> of each level (FabOnDisk names the file, byte offsets
follow from the box sizes alone, so no rank needs another's data), the per-fab minima / maxima are max-reduced over the ranks
(vdn_comm_allreduce_max, which is also the barrier), rank 0 writes the text files.  One shared directory, i.e. one node or a
shared file system."""
import os
import re

import numpy as np

from . import advance as adv
from . import boxlib as bl

FAB_DESC = "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"       # IEEE little-endian doubles, the "native" descriptor


def _es(x):
    """Fortran es27.17e3"""
    m, e = ("%.17E" % float(x)).split("E")
    return ("%sE%s%03d" % (m, e[0], int(e[1:]))).rjust(27)


def _boxstr(lo, hi, nodal, dm):
    j = lambda v: ",".join(str(int(x)) for x in v[:dm])   # noqa: E731
    return "((%s) (%s) (%s))" % (j(lo), j(hi), j(nodal))


# ---- one level -----------------------------------------------------------------------------------------------------------------------
def _write_level(dirname, prefix, boxes, nodal, fabs, dm, nc=None, owner=None, rank=0, reduce_max=None):
    """fabio_multifab_write_d: boxes = ALL (lo, hi) cell boxes of the level; fabs = arrays (nx, ny, nz, nc) of the valid points (incl. the
    nodal one) -- a list for every box, or {global box index: array} for the boxes this rank owns (owner = rank of every box)"""
    os.makedirs(dirname, exist_ok=True)
    if not isinstance(fabs, dict):
        fabs = dict(enumerate(fabs))
    if nc is None:
        nc = next(iter(fabs.values())).shape[3]
    owner = list(owner) if owner is not None else [0] * len(boxes)
    fname = lambda r: "%s_D_%05d" % (prefix, r)   # noqa: E731
    hdr = [("%s%s %d\n" % (FAB_DESC, _boxstr(lo, tuple(hi[d] + nodal[d] for d in range(3)), nodal, dm), nc)).encode() for lo, hi in boxes]
    offs, pos = [], {}
    for g, (lo, hi) in enumerate(boxes):                 # every rank's file holds its boxes in ascending global order
        offs.append(pos.get(owner[g], 0))
        pos[owner[g]] = offs[g] + len(hdr[g]) + 8 * nc * int(np.prod([hi[d] + nodal[d] - lo[d] + 1 for d in range(3)]))
    mm = np.full((2, len(boxes), max(nc, 1)), -np.inf)   # [0] = -min, [1] = max: one MAX reduction serves both
    mine = [g for g in range(len(boxes)) if owner[g] == rank]
    if mine:
        with open(os.path.join(dirname, fname(rank)), "wb") as f:
            for g in mine:
                lo, hi = boxes[g]
                a = fabs[g]
                assert a.shape == tuple(hi[d] + nodal[d] - lo[d] + 1 for d in range(3)) + (nc,), (a.shape, lo, hi, nodal, nc)
                assert f.tell() == offs[g]
                f.write(hdr[g])
                f.write(np.asfortranarray(a, dtype="<f8").tobytes(order="F"))
                mm[0, g, :nc] = [-float(a[..., c].min()) for c in range(nc)]
                mm[1, g, :nc] = [float(a[..., c].max()) for c in range(nc)]
    if reduce_max is not None:
        mm = reduce_max(mm.reshape(-1)).reshape(mm.shape)
    if rank != 0:
        return
    with open(os.path.join(dirname, prefix + "_H"), "w") as f:
        f.write("1\n0\n%d\n0\n" % nc)                                   # version, how, ncomp, nghost
        f.write("(%d 0\n" % len(boxes))
        for lo, hi in boxes:
            f.write(_boxstr(lo, tuple(hi[d] + nodal[d] for d in range(3)), nodal, dm) + "\n")
        f.write(")\n%d\n" % len(boxes))
        for g in range(len(boxes)):
            f.write("FabOnDisk: %s %d\n" % (fname(owner[g]), offs[g]))
        for rows in (-mm[0], mm[1]):
            f.write("\n%d,%d\n" % (len(boxes), nc))
            for r in rows:
                f.write("".join(_es(v) + "," for v in r[:nc]) + "\n")


def _read_level(dirname, prefix):
    with open(os.path.join(dirname, prefix + "_H")) as f:
        lines = f.read().split("\n")
    nc = int(lines[2])
    nb = int(lines[4].strip("(").split()[0])
    boxes, nodal = [], (0, 0, 0)
    for ln in lines[5:5 + nb]:
        g = [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\(([-\d,]+)\)", ln)]
        pad = lambda v, fill=0: tuple(v) + (fill,) * (3 - len(v))   # noqa: E731
        nodal = pad(g[2])
        boxes.append((pad(g[0]), tuple(h - n for h, n in zip(pad(g[1]), nodal))))
    k = 5 + nb + 2
    fabs = []
    for (lo, hi), ln in zip(boxes, lines[k:k + nb]):
        _, fname, off = ln.split()
        with open(os.path.join(dirname, fname), "rb") as f:
            f.seek(int(off))
            hdr = f.readline().decode()
            assert hdr.startswith(FAB_DESC), "unsupported FAB descriptor: " + hdr[:80]
            shp = tuple(hi[d] + nodal[d] - lo[d] + 1 for d in range(3)) + (nc,)
            a = np.frombuffer(f.read(8 * int(np.prod(shp))), dtype="<f8").reshape(shp, order="F")
        fabs.append(np.array(a, order="F"))
    return boxes, nodal, fabs


# ---- a hierarchy (fabio_ml_multifab_write_d / _read_d) ---------------------------------------------------------------------------------
def write_ml_multifab(dirname, levels, rr, dm=3, names=None, pd=None, prob_lo=None, prob_hi=None, time=0.0, dx=None, nc=None,
                      rank=0, reduce_max=None):
    """levels: per level dict(boxes=[(lo, hi)], nodal=(..), fabs=[arrays] or {box index: array}, owner=[rank of each box]);
    rr: refinement ratio between consecutive levels; pd: (lo, hi) of the level-0 domain; dx: level-0 mesh spacing.  The optional
    arguments default as in fabio (names Var-i, unit box).  Several ranks: everybody calls with its own fabs, see the module docstring."""
    nl = len(levels)
    if nc is None:
        f0 = levels[0]["fabs"]
        nc = (next(iter(f0.values())) if isinstance(f0, dict) else f0[0]).shape[3]
    names = list(names) if names else ["Var-%d" % (i + 1) for i in range(nc)]
    if pd is None:                                                          # bounding box of level 0
        los, his = zip(*levels[0]["boxes"])
        pd = (tuple(min(b[d] for b in los) for d in range(3)), tuple(max(b[d] for b in his) for d in range(3)))
    prob_lo = list(prob_lo) if prob_lo is not None else [0.0] * dm
    prob_hi = list(prob_hi) if prob_hi is not None else [float(pd[1][d] - pd[0][d] + 1) for d in range(dm)]
    dx = list(dx) if dx is not None else [(prob_hi[d] - prob_lo[d]) / (pd[1][d] - pd[0][d] + 1) for d in range(dm)]
    os.makedirs(dirname, exist_ok=True)
    for n, L in enumerate(levels):
        _write_level(os.path.join(dirname, "Level_%02d" % n), "Cell", L["boxes"], L.get("nodal", (0, 0, 0)), L["fabs"], dm, nc,
                     L.get("owner"), rank, reduce_max)
    if rank != 0:
        return
    with open(os.path.join(dirname, "Header"), "w") as f:
        f.write("NavierStokes-V1.1\n%d\n" % nc)
        for s in names:
            f.write(s.strip() + "\n")
        f.write("%d\n%s\n%d\n" % (dm, _es(time), nl - 1))
        f.write("".join(_es(v) for v in prob_lo[:dm]) + "\n" + "".join(_es(v) for v in prob_hi[:dm]) + "\n")
        f.write(" ".join(str(int(r)) for r in rr[:nl - 1]) + "\n")
        lo, hi = list(pd[0]), list(pd[1])
        doms = []
        for n in range(nl):
            doms.append(_boxstr(lo, hi, (0, 0, 0), dm))
            if n < nl - 1:
                lo, hi = [x * rr[n] for x in lo], [(x + 1) * rr[n] - 1 for x in hi]
        f.write(" ".join(doms) + "\n" + " ".join("0" for _ in range(nl)) + "\n")
        dxl = list(dx[:dm])
        dxs = []
        for n in range(nl):
            dxs.append(list(dxl))
            f.write("".join(_es(v) for v in dxl) + "\n")
            if n < nl - 1:
                dxl = [v / rr[n] for v in dxl]
        f.write("0\n0\n")
        for n, L in enumerate(levels):
            f.write("%d %d %s\n0\n" % (n, len(L["boxes"]), _es(time)))
            for lo, hi in L["boxes"]:
                for d in range(dm):
                    f.write(_es(prob_lo[d] + lo[d] * dxs[n][d]) + _es(prob_lo[d] + (hi[d] + 1) * dxs[n][d]) + "\n")
            f.write("Level_%02d/Cell\n" % n)


def read_ml_multifab(dirname):
    with open(os.path.join(dirname, "Header")) as f:
        ln = f.read().split("\n")
    nc = int(ln[1])
    names = [s.strip() for s in ln[2:2 + nc]]
    k = 2 + nc
    dm, time, nl = int(ln[k]), float(ln[k + 1]), int(ln[k + 2]) + 1
    prob_lo = [float(x) for x in ln[k + 3].split()]
    prob_hi = [float(x) for x in ln[k + 4].split()]
    rr = [int(x) for x in ln[k + 5].split()]
    dom = [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\(([-\d,]+)\)", ln[k + 6])]
    pad = lambda v: tuple(v) + (0,) * (3 - len(v))   # noqa: E731
    pd = (pad(dom[0]), pad(dom[1]))
    dx0 = [float(x) for x in ln[k + 8].split()]
    paths = [s for s in ln if re.fullmatch(r"Level_\d+/\w+", s.strip())]
    levels = []
    for p in paths[:nl]:
        sub, prefix = p.strip().split("/")
        boxes, nodal, fabs = _read_level(os.path.join(dirname, sub), prefix)
        levels.append(dict(boxes=boxes, nodal=nodal, fabs=fabs))
    return dict(names=names, dm=dm, time=time, nlevs=nl, prob_lo=prob_lo, prob_hi=prob_hi, rr=rr, pd=pd, dx=dx0, levels=levels)


# ---- the simulation objects of driver.py -----------------------------------------------------------------------------------------------
def _sim_levels(sim):
    """[(boxes, local indices)] per level of a Varden (flat lists) or VardenAMR (lists per level)"""
    if sim.boxes and isinstance(sim.boxes[0][0][0], int):
        return [(sim.boxes, sim.local)]
    return list(zip(sim.boxes, sim.local))


def _valid(mf, li):
    a = mf.to_numpy(li)
    g = mf.ng
    gz = g if bl._dm == 3 else 0
    return a[g:a.shape[0] - g, g:a.shape[1] - g, gz:a.shape[2] - gz] if g else a


def _owners(sim):
    return [sim.owner] if sim.owner and isinstance(sim.owner[0], int) else list(sim.owner)


def _gather(sim, mfs_per_level, nodal=(0, 0, 0)):
    out = []
    for n, (boxes, local) in enumerate(_sim_levels(sim)):
        fabs = {gi: np.concatenate([_valid(mf[n], li) for mf in mfs_per_level], axis=3) for li, gi in enumerate(local)}
        out.append(dict(boxes=list(boxes), nodal=nodal, fabs=fabs, owner=_owners(sim)[n]))
    return out


def _par(sim):
    """rank and the reduction of the writers"""
    nr = getattr(sim, "nranks", 1)
    return dict(rank=getattr(sim, "rank", 0), reduce_max=bl.comm_allreduce_max if nr > 1 else None)


def plot_names(dm, nscal):
    """src/varden.f90:73-87"""
    names = ["x_vel", "y_vel"] + (["z_vel"] if dm > 2 else []) + ["density"] + (["tracer"] if nscal > 1 else [])
    names += ["scalar_%d" % i for i in range(3, nscal + 1)]                 # the reference leaves these names blank
    return names + ["magvel", "vort", "gpx", "gpy"] + (["gpz"] if dm > 2 else [])


def _domain(sim):
    lv = _sim_levels(sim)
    n = sim.n if hasattr(sim, "n") else getattr(sim, "ncs", None) or (sim.nc,) * 3
    return ((0, 0, 0), tuple(int(x) - 1 for x in n)), len(lv)


def write_plotfile(sim, istep=None, base="plt", prob_lo=None, prob_hi=None):
    """write_plotfile(istep) of src/varden.f90:492-585: velocity, scalars, |u|, vorticity (src/makevort.f90, computed on the device),
    grad p -- 2 dm + nscal + 2 components on every level; returns the directory name plt<istep:05d>"""
    dm, ns = sim.dm, sim.nscal
    pd, nl = _domain(sim)
    ncomp = 2 * dm + ns + 2
    plot = [bl.MultiFab(sim.mla, n, ncomp, 0) for n in range(nl)]
    try:
        for n in range(nl):
            plot[n].copy_c(0, sim.uold[n], 0, dm)
            plot[n].copy_c(dm, sim.sold[n], 0, ns)
            adv.make_magvel(plot[n], dm + ns, sim.uold[n])
            adv.make_vorticity(plot[n], dm + ns + 1, sim.uold[n], sim.dx[n], sim.bct)
            plot[n].copy_c(dm + ns + 2, sim.gp[n], 0, dm)
        levels = _gather(sim, [plot])
    finally:
        for m in plot:
            m.destroy()
    name = "%s%05d" % (base, sim.istep if istep is None else istep)
    dx0 = list(sim.dx[0][:dm])
    hi = prob_hi if prob_hi is not None else [dx0[d] * (pd[1][d] + 1) for d in range(dm)]
    write_ml_multifab(name, levels, [2] * (nl - 1), dm, plot_names(dm, ns), pd, prob_lo or [0.0] * dm, hi, sim.time, dx0, nc=ncomp, **_par(sim))
    if getattr(sim, "rank", 0) == 0:
        write_job_info(name, sim, getattr(sim, "inputs_text", None), getattr(sim, "job_name", ""), getattr(sim, "inputs_file", ""))   # varden.f90:583
    return name


BC_NAMES = {-1: "periodic", 0: "interior", 11: "inlet", 12: "outlet", 13: "symmetry", 14: "slip wall", 15: "no slip wall"}


def write_job_info(dirname, sim, inputs_text=None, job_name="", inputs_file=""):
    """job_info in a plot directory (src/write_job_info.f90): job, output, grid and boundary-condition sections; the build section
    names this library instead of the Fortran tool chain; the run-time parameters are the namelist the run was started from"""
    import datetime
    bar = "=" * 79
    lv = _sim_levels(sim)
    pd, _ = _domain(sim)
    with open(os.path.join(dirname, "job_info"), "w") as f:
        f.write("%s\n Job Information\n%s\njob name:    %s\ninputs file: %s\n \n" % (bar, bar, job_name, inputs_file))
        f.write("number of MPI processes %6d\nnumber of threads       %6d\n \n \n" % (getattr(sim, "nranks", 1), 1))
        now = datetime.datetime.now()
        f.write("%s\n Plotfile Information\n%s\noutput date:              %s\noutput time:              %s\noutput dir:               %s\n \n \n"
                % (bar, bar, now.strftime("%Y-%m-%d"), now.strftime("%H:%M:%S"), os.getcwd()))
        f.write("%s\n Build Information\n%s\nvarden_amd (MI355X, HIP): %s\n \n \n" % (bar, bar, os.path.dirname(os.path.abspath(__file__))))
        f.write("%s\n Grid Information\n%s\n" % (bar, bar))
        for n, (boxes, _) in enumerate(lv):
            ext = [(pd[1][d] + 1) << n for d in range(sim.dm)]
            f.write(" level: %d\n    number of boxes = %d\n    maximum zones   = %s\n" % (n + 1, len(boxes), " ".join(str(e) for e in ext)))
        f.write(" \n Boundary Conditions\n")
        for d in range(sim.dm):
            f.write("   -%s: %s\n   +%s: %s\n \n" % ("xyz"[d], BC_NAMES.get(sim.phys[d][0], str(sim.phys[d][0])), "xyz"[d], BC_NAMES.get(sim.phys[d][1], str(sim.phys[d][1]))))
        f.write(" \n%s\n Runtime Parameter Information\n%s\n%s\n" % (bar, bar, (inputs_text or "").strip()))


def write_grids(grids_file_name, sim, nstep):
    """write_grids of src/varden.f90:621-662: the box lists of all levels appended to the grids file (read back by fixed_grids runs)"""
    lv = _sim_levels(sim)
    pd, _ = _domain(sim)
    dm = sim.dm
    fmt = lambda lo, hi: "((%s) (%s) (%s))" % (", ".join(str(int(x)) for x in lo[:dm]), ", ".join(str(int(x)) for x in hi[:dm]), ",".join("0" for _ in range(dm)))   # noqa: E731
    if getattr(sim, "rank", 0) != 0:
        return
    with open(grids_file_name, "a") as f:
        f.write("At step %5d:\n%2d\n" % (nstep, len(lv)))
        for n, (boxes, _) in enumerate(lv):
            dlo, dhi = [x << n for x in pd[0]], [((x + 1) << n) - 1 for x in pd[1]]
            f.write("   %s %4d\n" % (fmt(dlo, dhi), len(boxes)))
            for lo, hi in boxes:
                f.write("      %s \n" % fmt(lo, hi))
        f.write(" \n")


def read_grids(grids_file_name):
    """the box lists of a fixed_grids file (FBoxLib read_a_hgproj_grid, src/initialize.f90:112): number of levels, then per level the
    domain box with the number of boxes and the boxes -- the block write_grids appends ("At step" lines are skipped, the first block is read)"""
    rows = [ln.strip() for ln in open(grids_file_name) if ln.strip() and not ln.startswith("At step")]
    nlev = int(rows[0])
    pad = lambda v: tuple(v) + (0,) * (3 - len(v))   # noqa: E731
    box = lambda ln: [pad(tuple(int(x) for x in t.split(","))) for t in re.findall(r"\(([-\d, ]+)\)", ln)]   # noqa: E731
    k, levels, domains = 1, [], []
    for _ in range(nlev):
        g = box(rows[k])
        nb = int(rows[k].split()[-1])
        domains.append((g[0], g[1]))
        levels.append([tuple(box(rows[k + 1 + i])[:2]) for i in range(nb)])
        k += 1 + nb
    return domains, levels


def write_checkfile(sim, istep=None, base="chk"):
    """write_checkfile of src/varden.f90:587-610 + checkpoint_write (src/checkpoint.f90:15-86): State = (uold, sold, gp) valid cells,
    Pressure = nodal p, Header = namelist &chkpoint (time, dt, nlevs) followed by the refinement ratios"""
    name = "%s%05d" % (base, sim.istep if istep is None else istep)
    pd, nl = _domain(sim)
    os.makedirs(name, exist_ok=True)
    write_ml_multifab(os.path.join(name, "State"), _gather(sim, [sim.uold, sim.sold, sim.gp]), [2] * (nl - 1), sim.dm, pd=pd,
                      nc=2 * sim.dm + sim.nscal, **_par(sim))
    nd = (1, 1, 1) if sim.dm == 3 else (1, 1, 0)
    write_ml_multifab(os.path.join(name, "Pressure"), _gather(sim, [sim.p], nd), [2] * (nl - 1), sim.dm, pd=pd, nc=1, **_par(sim))
    if getattr(sim, "rank", 0) != 0:
        return name
    with open(os.path.join(name, "Header"), "w") as f:
        f.write("&CHKPOINT\n TIME=%s,\n DT=%s,\n NLEVS=%d,\n /\n" % (_es(sim.time).strip(), _es(sim.dt).strip(), nl))
        for _ in range(nl - 1):
            f.write("%12d\n" % 2)
    return name


def read_checkfile(name):
    """checkpoint_read (src/checkpoint.f90:88-145) + fill_restart_data (src/restart.f90:17-50): the box lists come from the file"""
    with open(os.path.join(name, "Header")) as f:
        text = f.read()
    nml = {k.lower(): v for k, v in re.findall(r"(\w+)\s*=\s*([-+\w.]+)", text)}
    nl = int(nml["nlevs"])
    tail = text[text.index("/") + 1:].split()
    state, press = read_ml_multifab(os.path.join(name, "State")), read_ml_multifab(os.path.join(name, "Pressure"))
    assert state["nlevs"] == nl and press["nlevs"] == nl
    return dict(nlevs=nl, time=float(nml["time"].lower().replace("d", "e")), dt=float(nml["dt"].lower().replace("d", "e")),
                rr=[int(x) for x in tail[:nl - 1]], dm=state["dm"], pd=state["pd"],
                boxes=[L["boxes"] for L in state["levels"]], state=[L["fabs"] for L in state["levels"]],
                pressure=[L["fabs"] for L in press["levels"]])


def load_restart(sim, chk):
    """initialize_from_restart (src/initialize.f90:52-57): uold, sold, gp, p <- the checkpoint; the sim was built on chk['boxes']"""
    dm, ns = sim.dm, sim.nscal
    for n, (boxes, local) in enumerate(_sim_levels(sim)):
        assert [tuple(map(tuple, b)) for b in boxes] == [tuple(map(tuple, b)) for b in chk["boxes"][n]], "restart: box lists differ"
        for li, gi in enumerate(local):
            st, pr = chk["state"][n][gi], chk["pressure"][n][gi]
            for mf, c0, nc in ((sim.uold[n], 0, dm), (sim.sold[n], dm, ns), (sim.gp[n], dm + ns, dm), (sim.p[n], None, 1)):
                a = mf.to_numpy(li)
                g = mf.ng
                gz = g if dm == 3 else 0
                v = a[g:a.shape[0] - g, g:a.shape[1] - g, gz:a.shape[2] - gz]
                v[...] = pr if c0 is None else st[..., c0:c0 + nc]
                mf.from_numpy(a, li)
    sim.time, sim.dt = chk["time"], chk["dt"]
