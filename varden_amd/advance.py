"""The reference's hot-path entry points, same names and argument order, over the C-ABI.

    advance_timestep   reference src/advance_timestep.f90:26-44
    estdt              reference src/estdt.f90:15
    hgproject          reference src/hgproject.f90:17-18
    macproject         reference src/macproject.f90:20
    velpred / mkflux / update / mkvelforce / mkscalforce / make_at_halftime : the per-kernel modules

Multifab arguments are lists with one :class:`MultiFab` per level, as in the Fortran (``sold(:)``).
"""
import ctypes as C

from . import capi
from .boxlib import handle_array
from .capi import check


def _dx(dx):
    flat = [float(v) for lev in dx for v in lev] if hasattr(dx[0], "__len__") else [float(v) for v in dx]
    return (C.c_double * len(flat))(*flat)


def advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force,
                     the_bc_tower, dt, time, dx, press_comp, proj_type):
    lib = capi.load()
    check(lib.vdn_advance_timestep(istep, mla.h, handle_array(sold), handle_array(uold), handle_array(snew),
                                   handle_array(unew), handle_array(gp), handle_array(p),
                                   handle_array(ext_vel_force), handle_array(ext_scal_force), the_bc_tower.h,
                                   float(dt), float(time), _dx(dx), press_comp, proj_type))


def estdt(lev, u, s, gp, ext_vel_force, dx, dtold):
    """returns dt (the reference's intent(out) argument)"""
    out = C.c_double()
    check(capi.load().vdn_estdt(lev, u.h, s.h, gp.h, ext_vel_force.h, _dx(dx), float(dtold), C.byref(out)))
    return out.value


def hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, the_bc_tower, press_comp):
    check(capi.load().vdn_hgproject(proj_type, mla.h, handle_array(unew), handle_array(uold), handle_array(rhohalf),
                                    handle_array(p), handle_array(gp), _dx(dx), float(dt), the_bc_tower.h, press_comp))


def macproject(mla, umac, rho, mac_rhs, dx, the_bc_tower, bc_comp):
    """umac: list per level of [umac, vmac, wmac]"""
    flat = [m for lev in umac for m in lev]
    check(capi.load().vdn_macproject(mla.h, handle_array(flat), handle_array(rho), handle_array(mac_rhs), _dx(dx),
                                     the_bc_tower.h, bc_comp))


def last_step_timing():
    t = (C.c_double * 5)()
    capi.load().vdn_last_step_timing(t)
    return dict(scalar=t[0], velocity=t[1], mac=t[2], hg=t[3], total=t[4])


def last_solver_stats(which):
    cyc, r0, r = C.c_int(), C.c_double(), C.c_double()
    capi.load().vdn_last_solver_stats(0 if which == "mac" else 1, C.byref(cyc), C.byref(r0), C.byref(r))
    return cyc.value, r0.value, r.value


# ---- per-kernel modules (single level) ------------------------------------------------------------
def _iv(x):
    return (C.c_int * len(x))(*[int(v) for v in x])


def slope(s, slope_mf, dir, bccomp, bct):
    check(capi.load().vdn_k_slope(s.h, slope_mf.h, dir, bccomp, bct.h))


def velpred(u, umac, force, dx, dt, bct):
    check(capi.load().vdn_k_velpred(u.h, handle_array(umac), force.h, _dx(dx), float(dt), bct.h))


def mkflux(s, sedge, flux, umac, force, mac_rhs, dx, dt, bct, is_vel, is_conservative):
    check(capi.load().vdn_k_mkflux(s.h, handle_array(sedge), handle_array(flux), handle_array(umac), force.h, mac_rhs.h,
                                   _dx(dx), float(dt), bct.h, 1 if is_vel else 0, _iv(is_conservative)))


def update(sold, umac, sedge, flux, force, snew, dx, dt, is_vel, is_cons, bct):
    check(capi.load().vdn_k_update(sold.h, handle_array(umac), handle_array(sedge), handle_array(flux), force.h, snew.h,
                                   _dx(dx), float(dt), 1 if is_vel else 0, _iv(is_cons), bct.h))


def mkvelforce(vel_force, ext_vel_force, s, gp, lapu, visc_fac, bct):
    check(capi.load().vdn_k_mkvelforce(vel_force.h, ext_vel_force.h, s.h, gp.h, lapu.h if lapu else None, float(visc_fac), bct.h))


def mkscalforce(scal_force, ext_scal_force, laps, diff_fac, bct):
    check(capi.load().vdn_k_mkscalforce(scal_force.h, ext_scal_force.h, laps.h if laps else None, float(diff_fac), bct.h))


def make_at_halftime(rhohalf, sold, snew, in_comp, out_comp, bct):
    check(capi.load().vdn_k_make_at_halftime(rhohalf.h, sold.h, snew.h, in_comp, out_comp, bct.h))


def cc_solve(rh, phi, beta, dx, bc, rel_eps, abs_eps=-1.0, max_iter=100):
    cyc, r0, r = C.c_int(), C.c_double(), C.c_double()
    flat = _iv([bc[d][s] for d in range(3) for s in range(2)])
    check(capi.load().vdn_cc_solve(rh.h, phi.h, handle_array(beta), _dx(dx), flat, rel_eps, abs_eps, max_iter,
                                   C.byref(cyc), C.byref(r0), C.byref(r)))
    return cyc.value, r0.value, r.value


def cc_smooth(rh, phi, beta, dx, bc, nsweeps):
    flat = _iv([bc[d][s] for d in range(3) for s in range(2)])
    check(capi.load().vdn_cc_smooth(rh.h, phi.h, handle_array(beta), _dx(dx), flat, nsweeps))


def nd_solve(rh, phi, coeffs, u, dx, bc, rel_eps, abs_eps=-1.0, max_iter=100):
    cyc, r0, r = C.c_int(), C.c_double(), C.c_double()
    flat = _iv([bc[d][s] for d in range(3) for s in range(2)])
    check(capi.load().vdn_nd_solve(rh.h, phi.h, coeffs.h, u.h if u else None, _dx(dx), flat, rel_eps, abs_eps, max_iter,
                                   C.byref(cyc), C.byref(r0), C.byref(r)))
    return cyc.value, r0.value, r.value


def bench_cc_smoother(rh, phi, beta, dx, bc, nlaunch, rho=None):
    """rho: the density behind beta -- the pass macproject runs (coefficients recomputed from rho); None: the stored-coefficient pass"""
    ms, cells = C.c_double(), C.c_long()
    flat = _iv([bc[d][s] for d in range(3) for s in range(2)])
    check(capi.load().vdn_bench_cc_smoother(rh.h, phi.h, handle_array(beta), rho.h if rho is not None else None, _dx(dx), flat, nlaunch,
                                            C.byref(ms), C.byref(cells)))
    return ms.value, cells.value


def bench_cc_smoother_in_solve(rh, phi, beta, dx, bc, nsweeps, nlaunch, rho):
    """the colour passes as a solve launches them (sweeps time-skewed over plane slabs): (ms per pass over the whole level, cells); cells = 0: no slab schedule here"""
    ms, cells = C.c_double(), C.c_long()
    flat = _iv([bc[d][s] for d in range(3) for s in range(2)])
    check(capi.load().vdn_bench_cc_smoother_in_solve(rh.h, phi.h, handle_array(beta), rho.h, _dx(dx), flat, nsweeps, nlaunch, C.byref(ms), C.byref(cells)))
    return ms.value, cells.value


# ---- multi-level operators (two levels) -----------------------------------------------------------------------------------
def ml_cc_restriction(crse, fine, icomp=0, nc=None):
    check(capi.load().vdn_ml_cc_restriction(crse.h, fine.h, icomp, crse.nc if nc is None else nc))


def ml_edge_restriction(crse, fine, dir):
    check(capi.load().vdn_ml_edge_restriction(crse.h, fine.h, dir))


def fill_ghost_cells(fine, crse, icomp=0, nc=None):
    check(capi.load().vdn_multifab_fill_ghost_cells(fine.h, crse.h, icomp, fine.nc if nc is None else nc))


def create_umac_grown(fine, crse, dir):
    check(capi.load().vdn_create_umac_grown(fine.h, crse.h, dir))


def ml_restrict_and_fill(mfs, icomp, bcomp, nc, bct, same_boundary=False):
    check(capi.load().vdn_ml_restrict_and_fill(len(mfs), handle_array(mfs), icomp, bcomp, nc, 1 if same_boundary else 0, bct.h))


def make_new_grids(s, lev, buf_wid=2, nest=2, min_eff=0.9, min_width=4, blocking=4, max_grid_size=256, maxboxes=4096):
    """tag_boxes (src/tag_boxes.f90) on component 0 of `s` (the state of level `lev`, 1-based) + FBoxLib's make_new_grids
    (src/initialize.f90:247-248): the boxes of level lev+1 in that level's index space ([] = nothing tagged), and the tag count"""
    boxes = (capi.Box * maxboxes)()
    nb, nt = C.c_int(), C.c_long()
    check(capi.load().vdn_make_new_grids(s.h, lev, buf_wid, nest, min_eff, min_width, blocking, max_grid_size, maxboxes, boxes,
                                         C.byref(nb), C.byref(nt)))
    return [(tuple(boxes[i].lo), tuple(boxes[i].hi)) for i in range(nb.value)], nt.value


def tag_boxes(s, lev):
    """tag_boxes of src/tag_boxes.f90 on component 0 of `s` (level `lev`, 1-based): uint8 array over the level's domain (x first)"""
    import numpy as np
    lo, hi = s.mla.pd[s.lev]
    n = tuple(hi[d] - lo[d] + 1 for d in range(3))
    tags = np.zeros(n, dtype=np.uint8, order="F")
    check(capi.load().vdn_tag_boxes(s.h, lev, tags.ctypes.data_as(C.POINTER(C.c_ubyte))))
    return tags


def fillpatch(fine, crse, icomp, nc):
    """fillpatch(fine, crse, 0, ...) of src/regrid.f90:311-325: valid cells of a new fine level from the coarser one"""
    check(capi.load().vdn_fillpatch(fine.h, crse.h, icomp, nc))


def make_vorticity(vort, comp, u, dx, bct):
    """make_vorticity(vort, comp, u, dx, bc) of src/makevort.f90:16-57 (comp 0-based; fills the ghost cells of u)"""
    d = (C.c_double * 3)(*(list(dx) + [1.0] * 3)[:3])
    check(capi.load().vdn_make_vorticity(vort.h, comp, u.h, d, bct.h))


def make_magvel(magvel, comp, u):
    """make_magvel(magvel, comp, u) of src/makevort.f90:59-91"""
    check(capi.load().vdn_make_magvel(magvel.h, comp, u.h))


def ml_nodal_prolongation(fine, crse):
    check(capi.load().vdn_ml_nodal_prolongation(fine.h, crse.h))


def copy_layouts(dst, dcomp, src, scomp, nc):
    """multifab_copy_c between two multifabs of one level whose box lists differ (src/regrid.f90:333-337)"""
    check(capi.load().vdn_multifab_copy_layouts(dst.h, dcomp, src.h, scomp, nc))
