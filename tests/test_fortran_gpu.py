"""north_star: "Host code stays Fortran (ISO_C_BINDING)".  tests/fortran/varden_drv -- a flang-built driver that calls the hot path
through varden_amd_mod.f90, the module that keeps the reference's names and argument lists (advance_timestep.f90:26-44, estdt.f90:15,
hgproject.f90:17, varden.f90:291-328) -- runs as a fresh child process and must reproduce, step for step, what the Python mirror of the same
flow computes through ctypes: both are thin hosts over ONE C-ABI, so time, dt and max|u| agree to the last printed digit (17 significant
digits; the test asks for 1e-12).  nlevs = 2 exercises the multi-level conventions from Fortran: multifab arrays indexed by level,
dx(level, dir), umac[lev*3 + dir] inside, ml_restrict_and_fill with 1-based component arguments."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "varden_amd", "fortran")           # the product's modules and varden_main
XDIR = os.path.join(ROOT, "tests", "fortran")                # the boundary fixtures varden_drv / varden_loop (test infrastructure)
DRV = os.path.join(XDIR, "varden_drv")
WALLS = [[15, 15]] * 3


def _run_fortran(n, nsteps, nlevs):
    if not os.path.exists(DRV):                                   # built by __graft_entry__.build(); build here when flang is at hand
        if shutil.which("amdflang") is None and not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("no flang on this box and no prebuilt varden_drv")
        subprocess.check_call(["make", "-s", "-C", XDIR])
    out = subprocess.run([DRV, str(n), str(nsteps), str(nlevs)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rows = []
    for ln in out.stdout.splitlines():
        m = re.match(r"\s*step\s+(\d+)\s+time\s+(\S+)\s+dt\s+(\S+)\s+\|u\|max\s+(\S+)", ln)
        if m:
            rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))))
    assert len(rows) == nsteps, out.stdout
    return rows


@pytest.mark.parametrize("nlevs,n", [(1, 32), (2, 16)])
def test_fortran_driver_matches_the_python_mirror(gpu, nlevs, n):
    from varden_amd import driver
    from varden_amd.capi import default_params
    nsteps = 5
    frows = _run_fortran(n, nsteps, nlevs)
    if nlevs == 1:
        G = driver.Varden(n, WALLS, default_params(cflfac=0.9), prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1)
    else:
        fine = [((n // 2,) * 3, (3 * n // 2 - 1,) * 3)]
        G = driver.VardenAMR(n, fine, WALLS, params=default_params(cflfac=0.9), prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1,
                             do_initial_projection=1)
    for istep, ftime, fdt, fumax in frows:
        G.step()
        unew = G.unew if isinstance(G.unew, list) else [G.unew]
        umax = max(m.norm_inf() for m in unew[:nlevs])
        assert abs(G.dt - fdt) <= 1e-12 * fdt, (istep, G.dt, fdt)
        assert abs(G.time - ftime) <= 1e-12 * ftime, (istep, G.time, ftime)
        assert abs(umax - fumax) <= 1e-12 * fumax, (istep, umax, fumax)
    assert frows[-1][3] > 0.0 and np.isfinite(frows[-1][3])
    G.close()


LOOP = os.path.join(XDIR, "varden_loop")


@pytest.mark.parametrize("nlevs,n", [(1, 32), (2, 16)])
def test_the_reference_call_syntax_drops_onto_the_library(gpu, nlevs, n):
    """VERDICT r4 item 2 / BASELINE.json north_star ("the Fortran driver drops onto it unchanged"): varden_loop.f90 spells the driver flow of
    src/varden.f90 with the reference's OWN `use` lines (multifab_module, ml_layout_module, define_bc_module, advance_module, estdt_module, ...) and
    call syntax -- multifab_build(mf, mla%la(n), nc, ng[, nodal]), multifab_physbc(s, 1, 1, dm, the_bc_tower%bc_tower_array(n)),
    multifab_fill_ghost_cells(fine, crse, ng, mla%mba%rr(n-1,:), bc(n-1), bc(n), 1, 1, dm), multifab_copy_c(..., ng=unew(n)%ng), setval(..., all=.true.),
    ml_restrict_and_fill(nlevs, mf, mla%mba%rr, bc_tower_array, bcomp=...) -- over the modules of varden_boxlib.f90.  It must print the step lines of
    varden_drv (the same flow on the flat varden_amd module) character for character: time, dt and max|u| with 17 significant digits."""
    for exe in (DRV, LOOP):
        if not os.path.exists(exe):
            if shutil.which("amdflang") is None and not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
                pytest.skip("no flang on this box and no prebuilt Fortran executables")
            subprocess.check_call(["make", "-s", "-C", XDIR])
    lines = []
    for exe in (DRV, LOOP):
        out = subprocess.run([exe, str(n), "4", str(nlevs)], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        lines.append([ln for ln in out.stdout.splitlines() if re.match(r"\s*step\s+\d+", ln)])
    assert len(lines[0]) == 4 and lines[0] == lines[1], "varden_drv:\n%s\nvarden_loop:\n%s" % ("\n".join(lines[0]), "\n".join(lines[1]))


MAIN = os.path.join(FDIR, "varden_main")


@pytest.mark.parametrize("name,nsteps,nregrids", [("inputs_3d-regt", 6, 3), ("inputs_bubble_3d", 5, 3), ("inputs_advect_3d", 4, 2), ("inputs_RayleighTaylor_3d", 4, 4),
                                                  ("inputs_vortextube_3d", 3, 0),
                                                  # round 6: the four 2-D inputs (all adaptive) as z-uniform copies on the 3-D machinery (varden_main.f90: extruded)
                                                  ("inputs_2d-regt", 6, 3), ("inputs_bubble_2d", 5, 3), ("inputs_advect_2d", 4, 2), ("inputs_RayleighTaylor_2d", 4, 4)])
def test_fortran_main_runs_the_regression_inputs_like_the_python_mirror(gpu, tmp_path, name, nsteps, nregrids):
    """varden_main.f90 -- the flow of src/varden.f90 in Fortran: &PROBIN namelist, level 0 cut by max_grid_size, refined levels
    from tag_boxes + make_new_grids, start-up sequence, time loop with regrid every regrid_int steps through fillpatch / ml_nodal_prolongation /
    copies between box lists (src/regrid.f90:17-263), viscous solves -- on EVERY 3-D inputs file of exec/test (three-level regression case 64^3; the
    two-level bubble; inflow / outflow with three levels, prob_type 2; Rayleigh-Taylor, periodic x and y, regrid every step, prob_type 3; the triply
    periodic vortex tube on one level, prob_type 4), a few steps each, against varden_amd/inputs.py: run on the same file: the same boxes on every
    level, and time, dt, max|u| (the two hosts form the initial data with their own tanh / exp / sin: 1e-12 on time and dt, 1e-10 on max|u|)"""
    from varden_amd import inputs
    if not os.path.exists(MAIN):
        if shutil.which("amdflang") is None and not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("no flang on this box and no prebuilt varden_main")
        subprocess.check_call(["make", "-s", "-C", FDIR])
    path = os.path.join(ROOT, "tests", "golden", "inputs", name)
    out = subprocess.run([MAIN, path, str(nsteps)], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    frows = []
    for ln in out.stdout.splitlines():
        m = re.match(r"\s*step\s+(\d+)\s+time\s+(\S+)\s+dt\s+(\S+)\s+\|u\|max\s+(\S+)\s+levels\s+(\d+)\s+boxes\s+(.*)", ln)
        if m:
            frows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5)), [int(x) for x in m.group(6).split()]))
    assert len(frows) == nsteps, out.stdout[-3000:]
    regrids = int(re.search(r"regrids:\s+(\d+)", out.stdout).group(1))
    prows = []

    def report(G):
        amr = hasattr(G, "nlev")
        prows.append((G.istep, G.time, G.dt, max(m.norm_inf() for m in G.unew), G.nlev if amr else 1, [len(b) for b in G.boxes] if amr else [len(G.boxes)]))
    nl, G = inputs.run(open(path).read(), nsteps=nsteps, report=report, outdir=str(tmp_path))
    assert regrids == getattr(G, "nregrids", 0) == nregrids, (regrids, getattr(G, "nregrids", 0))
    for f, p in zip(frows, prows):
        assert f[0] == p[0] and f[4] == p[4] and f[5][:p[4]] == p[5], (f, p)
        assert abs(f[1] - p[1]) <= 1e-12 * p[1] and abs(f[2] - p[2]) <= 1e-12 * p[2] and abs(f[3] - p[3]) <= 1e-10 * p[3], (f, p)
    G.close()
