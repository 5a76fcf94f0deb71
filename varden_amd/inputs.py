"""Reads the reference's run-time inputs (a Fortran namelist &PROBIN, e.g. exec/test/inputs_bubble_3d) and drives the path the way
src/varden.f90 does: parameters (src/_parameters), grids (fixed single level, or tag_boxes + make_new_grids), start-up sequence,
time loop with estdt / regrid / advance_timestep, plot files every plot_int steps, checkpoints every chk_int steps, restart = n
continues from chk<n> (src/varden.f90:94-97, 207-229, 349-361)."""
import os
import re

from . import boxlib as bl
from . import plotfile
from .capi import default_params
from .driver import Varden, VardenAMR

# src/_parameters: the defaults that matter to the path
DEFAULTS = dict(dim_in=2, nscal=2, prob_type=1, grav=0.0, boussinesq=0, max_step=1, stop_time=-1.0, max_levs=1, max_grid_size=256,
                regrid_int=-1, amr_buf_width=-1, n_cellx=32, n_celly=32, n_cellz=32, prob_hi_x=1.0, prob_hi_y=1.0, prob_hi_z=1.0,
                init_iter=4, do_initial_projection=1, init_shrink=1.0, cflfac=0.8, max_dt_growth=1.1, visc_coef=0.0, diff_coef=0.0,
                diffusion_type=1, slope_order=4, use_minion=0, stencil_order=2, verbose=0, mg_verbose=0,
                bcx_lo=14, bcx_hi=14, bcy_lo=14, bcy_hi=14, bcz_lo=14, bcz_hi=14,
                fixed_dt=-1.0, plot_int=0, chk_int=0, restart=-1, plot_base_name="plt", check_base_name="chk", grids_file_name="", job_name="", fixed_grids="")


def parse_namelist(text):
    """key = value pairs of the first namelist group; Fortran literals (1.d0, .true., 'str'); indexed keys like u_bc(1,1) kept verbatim"""
    out = {}
    body = re.sub(r"!.*", "", text)
    for m in re.finditer(r"([A-Za-z_][\w]*(?:\(\s*\d+\s*,\s*\d+\s*\))?)\s*=\s*('[^'\n]*'|\"[^\"\n]*\"|[^\n,/]+)", body):
        key, val = m.group(1).replace(" ", ""), m.group(2).strip()
        low = val.lower()
        if low in (".true.", "t", ".t."):
            out[key] = 1
        elif low in (".false.", "f", ".f."):
            out[key] = 0
        elif val[:1] in "'\"":
            out[key] = val.strip("'\"")
        else:
            num = re.sub(r"[dD]", "e", val)
            try:
                out[key] = int(num)
            except ValueError:
                out[key] = float(num)
    return out


def build(text, device=0, max_grid_size_cap=None, outdir=".", extrude_nz=16, extrude_zbc=None):
    """the driver object for an inputs text: Varden (one level) or VardenAMR (max_levs > 1, grids from the tagged initial data);
    restart >= 0: grids and state from the checkpoint <outdir>/<check_base_name><restart:05d> (src/varden.f90:94-97)"""
    nl = dict(DEFAULTS)
    nl.update(parse_namelist(text))
    dm = int(nl["dim_in"])
    prm = default_params(dm=dm, nscal=int(nl["nscal"]), slope_order=int(nl["slope_order"]), use_minion=int(nl["use_minion"]),
                         boussinesq=int(nl["boussinesq"]), stencil_order=int(nl["stencil_order"]), diffusion_type=int(nl["diffusion_type"]),
                         verbose=int(nl["verbose"]), prob_type=int(nl["prob_type"]), visc_coef=float(nl["visc_coef"]),
                         diff_coef=float(nl["diff_coef"]), cflfac=float(nl["cflfac"]), max_dt_growth=float(nl["max_dt_growth"]))
    for name in ("u_bc", "v_bc", "w_bc", "rho_bc", "trac_bc"):           # inflow data, probin.template:21-23: name(direction, side)
        for key, v in nl.items():
            m = re.fullmatch(name + r"\((\d),(\d)\)", key)
            if m:
                getattr(prm, name)[int(m.group(1)) - 1][int(m.group(2)) - 1] = float(v)
    phys = [[int(nl["bc%s_lo" % a]), int(nl["bc%s_hi" % a])] for a in "xyz"[:dm]]
    n = tuple(int(nl["n_cell" + a]) for a in "xyz"[:dm])
    prob_hi = tuple(float(nl["prob_hi_" + a]) for a in "xyz"[:dm]) + (1.0,) * (3 - dm)
    mgs = int(nl["max_grid_size"]) if max_grid_size_cap is None else min(int(nl["max_grid_size"]), max_grid_size_cap)
    abw = max(int(nl["amr_buf_width"]), int(nl["regrid_int"]), 1)      # probin.template:147-154 (amr_buf_width >= regrid_int)
    common = dict(prob_type=int(nl["prob_type"]), grav=float(nl["grav"]), init_shrink=float(nl["init_shrink"]),
                  init_iter=int(nl["init_iter"]), do_initial_projection=int(nl["do_initial_projection"]), device=device,
                  fixed_dt=float(nl["fixed_dt"]), stop_time=float(nl["stop_time"]))
    decomp = tuple(max(1, -(-n[d] // mgs)) for d in range(dm)) + (1,) * (3 - dm)
    if int(nl["restart"]) >= 0:
        chk = plotfile.read_checkfile(os.path.join(outdir, "%s%05d" % (nl["check_base_name"], int(nl["restart"]))))
        rs = dict(restart=chk, restart_step=int(nl["restart"]))
        if chk["nlevs"] == 1:
            return nl, Varden(n, phys, prm, prob_hi=prob_hi, decomp=decomp, **common, **rs)
        if dm == 2:
            raise NotImplementedError("restart of a 2-D hierarchy (an extruded copy, DESIGN section 13): not in this round -- its checkpoint is the 3-D copy's")
        return nl, VardenAMR(n[0], chk["boxes"][1], phys, params=prm, finer_levels=chk["boxes"][2:], base_boxes=chk["boxes"][0],
                             regrid_int=int(nl["regrid_int"]), amr_buf_width=abw, max_levs=int(nl["max_levs"]), max_grid_size=mgs, **common, **rs)
    if nl["fixed_grids"]:                                   # initialize_with_fixed_grids, src/initialize.f90:93-150
        if dm == 2:
            raise NotImplementedError("fixed_grids with dim_in = 2: not in this round (adaptive 2-D hierarchies run as extruded copies)")
        if dm != 3 or len(set(n)) != 1 or any(p != 1.0 for p in prob_hi):
            raise NotImplementedError("hierarchies: 3-D, cubic unit domain in this round")
        domains, boxes = plotfile.read_grids(os.path.join(outdir, str(nl["fixed_grids"])))
        assert domains[0] == ((0, 0, 0), tuple(x - 1 for x in n)), "fixed_grids: level-0 domain differs from n_cell"
        if len(boxes) == 1:
            raise NotImplementedError("fixed_grids with one level: use max_grid_size")
        return nl, VardenAMR(n[0], boxes[1], phys, params=prm, finer_levels=boxes[2:], base_boxes=boxes[0], regrid_int=int(nl["regrid_int"]), amr_buf_width=abw,
                             max_levs=max(int(nl["max_levs"]), len(boxes)), max_grid_size=mgs, **common)
    if int(nl["max_levs"]) <= 1:
        return nl, Varden(n, phys, prm, prob_hi=prob_hi, decomp=decomp, **common)
    if dm == 2:
        # the 2-D inputs of exec/test (all four adaptive): the hierarchy runs as the z-uniform, z-periodic 3-D copy of the problem (driver.VardenAMR: extrude2d) --
        # plane k = 0 of every field is the 2-D answer, plot files are those of the 3-D copy
        if prob_hi[0] != 1.0 or abs(prob_hi[1] / n[1] - prob_hi[0] / n[0]) > 1e-15:
            raise NotImplementedError("2-D hierarchies: prob_hi_x = 1 and square cells")
        nz = int(extrude_nz)                                  # cells of level 0 along the periodic z of the copy (a multiple of the blocking factor)
        prm3 = default_params(dm=3, nscal=int(nl["nscal"]), slope_order=int(nl["slope_order"]), use_minion=int(nl["use_minion"]),
                              boussinesq=int(nl["boussinesq"]), stencil_order=int(nl["stencil_order"]), diffusion_type=int(nl["diffusion_type"]),
                              verbose=int(nl["verbose"]), prob_type=int(nl["prob_type"]), visc_coef=float(nl["visc_coef"]),
                              diff_coef=float(nl["diff_coef"]), cflfac=float(nl["cflfac"]), max_dt_growth=float(nl["max_dt_growth"]))
        for name in ("u_bc", "v_bc", "rho_bc", "trac_bc"):
            for d in range(2):
                for sd in range(2):
                    getattr(prm3, name)[d][sd] = getattr(prm, name)[d][sd]
        base = None
        if any(dc > 1 for dc in decomp[:2]) or nz > mgs:
            bs = [n[0] // decomp[0], n[1] // decomp[1], min(nz, mgs)]
            base = [((kx * bs[0], ky * bs[1], kz * bs[2]), ((kx + 1) * bs[0] - 1, (ky + 1) * bs[1] - 1, (kz + 1) * bs[2] - 1))
                    for kz in range(nz // bs[2]) for ky in range(decomp[1]) for kx in range(decomp[0])]
        levels = VardenAMR.tagged_grids(n, phys, prm3, prob_type=int(nl["prob_type"]), max_levs=int(nl["max_levs"]), buf_wid=abw, max_grid_size=mgs, device=device,
                                        base_boxes=base, extrude2d=nz)
        if not levels:
            return nl, Varden(n, phys, prm, prob_hi=prob_hi, decomp=decomp, **common)
        return nl, VardenAMR(n, levels[0], phys, params=prm3, finer_levels=levels[1:], regrid_int=int(nl["regrid_int"]), amr_buf_width=abw,
                             max_levs=int(nl["max_levs"]), max_grid_size=mgs, base_boxes=base, extrude2d=nz, extrude_zbc=extrude_zbc, **common)
    if len(set(n)) != 1 or any(p != 1.0 for p in prob_hi):
        raise NotImplementedError("adaptive hierarchies: cubic unit domain in this round")
    # level 0 is cut by max_grid_size like every other level (boxarray_maxsize, src/initialize.f90:204-206)
    base = None
    if any(dc > 1 for dc in decomp):
        bs = [n[d] // decomp[d] for d in range(3)]
        base = [((kx * bs[0], ky * bs[1], kz * bs[2]), ((kx + 1) * bs[0] - 1, (ky + 1) * bs[1] - 1, (kz + 1) * bs[2] - 1))
                for kz in range(decomp[2]) for ky in range(decomp[1]) for kx in range(decomp[0])]
    levels = VardenAMR.tagged_grids(n[0], phys, prm, prob_type=int(nl["prob_type"]), max_levs=int(nl["max_levs"]),
                                    buf_wid=abw, max_grid_size=mgs, device=device, base_boxes=base)
    if not levels:
        return nl, Varden(n, phys, prm, prob_hi=prob_hi, decomp=decomp, **common)
    return nl, VardenAMR(n[0], levels[0], phys, params=prm, finer_levels=levels[1:], regrid_int=int(nl["regrid_int"]), amr_buf_width=abw,
                         max_levs=int(nl["max_levs"]), max_grid_size=mgs, base_boxes=base, **common)


def run(text, nsteps=None, report=print, device=0, outdir=".", extrude_nz=16):
    """the time loop of src/varden.f90:237-371 for max_step steps (or until stop_time); plot / checkpoint files at step 0 of a fresh
    run and after every plot_int-th / chk_int-th step (:207-221, :349-361) under outdir"""
    nl, G = build(text, device=device, outdir=outdir, extrude_nz=extrude_nz)
    max_step = int(nl["max_step"]) if nsteps is None else nsteps
    stop_time = float(nl["stop_time"])
    plot_int, chk_int = int(nl["plot_int"]), int(nl["chk_int"])
    G.files_written = []
    G.inputs_text, G.job_name = text, str(nl["job_name"])
    grids_file = os.path.join(outdir, str(nl["grids_file_name"])) if nl["grids_file_name"] else None
    if grids_file and int(nl["restart"]) < 0 and hasattr(G, "nregrids"):
        plotfile.write_grids(grids_file, G, 0)                               # initialize.f90:340: the initial adaptive grids
    regrids_seen = [getattr(G, "nregrids", 0)]

    last = dict(plt=-1, chk=-1)

    def dump(final=False):
        """every plot_int-th / chk_int-th step; final: the last step once more if it has not been written (src/varden.f90:374-377)"""
        if plot_int > 0 and (G.istep % plot_int == 0 or final) and last["plt"] != G.istep:
            G.files_written.append(plotfile.write_plotfile(G, base=os.path.join(outdir, str(nl["plot_base_name"]))))
            last["plt"] = G.istep
        if chk_int > 0 and (G.istep % chk_int == 0 or final) and last["chk"] != G.istep:
            G.files_written.append(plotfile.write_checkfile(G, base=os.path.join(outdir, str(nl["check_base_name"]))))
            last["chk"] = G.istep

    if int(nl["restart"]) < 0:
        dump()
    while G.istep < max_step and (stop_time < 0 or G.time < stop_time):
        G.step()
        if grids_file and getattr(G, "nregrids", 0) != regrids_seen[0]:      # varden.f90:263-264
            regrids_seen[0] = G.nregrids
            plotfile.write_grids(grids_file, G, G.istep)
        if report:
            report(G)
        dump()
    if G.istep > (int(nl["restart"]) if int(nl["restart"]) >= 0 else 0):
        dump(final=True)
    return nl, G
