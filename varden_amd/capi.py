"""ctypes binding of the C-ABI declared in include/varden_amd.h.

The shared library (varden_amd/csrc/libvarden_amd.so) is built by ``__graft_entry__.build()``
(hipcc --offload-arch=gfx950).  There is NO CPU fallback: if the library is missing, or if it
cannot initialise a GPU, every product entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDN_LIB_FLAVOUR=testing: the suite's build of the same objects with the launch-form switches (VDN_*) and the test-transport seam compiled in
# (csrc/Makefile); anything else: the product, which reads no environment variable
FLAVOUR = "testing" if os.environ.get("VDN_LIB_FLAVOUR") == "testing" else "release"
LIB_PATH = os.path.join(_HERE, "csrc", "libvarden_amd_testing.so" if FLAVOUR == "testing" else "libvarden_amd.so")


class Params(C.Structure):
    """mirror of ``vdn_params`` (include/varden_amd.h); defaults = reference src/_parameters."""
    _fields_ = [
        ("dm", C.c_int), ("nscal", C.c_int), ("slope_order", C.c_int), ("use_minion", C.c_int),
        ("boussinesq", C.c_int), ("stencil_order", C.c_int), ("diffusion_type", C.c_int),
        ("verbose", C.c_int), ("mg_verbose", C.c_int), ("prob_type", C.c_int),
        ("visc_coef", C.c_double), ("diff_coef", C.c_double), ("cflfac", C.c_double),
        ("max_dt_growth", C.c_double),
        ("u_bc", (C.c_double * 2) * 3), ("v_bc", (C.c_double * 2) * 3), ("w_bc", (C.c_double * 2) * 3),
        ("rho_bc", (C.c_double * 2) * 3), ("trac_bc", (C.c_double * 2) * 3),
        ("mg_nu1", C.c_int), ("mg_nu2", C.c_int), ("mg_nub", C.c_int), ("mg_max_iter", C.c_int),
        ("hg_max_iter", C.c_int), ("hg_nu1", C.c_int), ("hg_nu2", C.c_int), ("hg_nub", C.c_int),
        ("hg_omega", C.c_double), ("mac_rel_eps", C.c_double), ("hg_rel_eps", C.c_double),
        ("abort_on_max_iter", C.c_int), ("hg_fmg", C.c_int), ("mac_fmg", C.c_int), ("hg_omega_pre1", C.c_double), ("hg_omega_pre2", C.c_double),
        ("hg_omega_fac1", C.c_double), ("hg_omega_fac2", C.c_double), ("hg_omega_fac3", C.c_double), ("mg_predict", C.c_int),
    ]


def default_params(**kw):
    """reference defaults (src/_parameters:10-92) + our multigrid controls; keep in sync with
    vdn_params_default() in varden_amd/csrc/runtime.hip (tests/test_capi_cpu.py checks it)."""
    p = Params()
    p.dm = 3; p.nscal = 2; p.slope_order = 4; p.use_minion = 0; p.boussinesq = 0
    p.stencil_order = 2; p.diffusion_type = 1; p.verbose = 0; p.mg_verbose = 0; p.prob_type = 1
    p.visc_coef = 0.0; p.diff_coef = 0.0; p.cflfac = 0.8; p.max_dt_growth = 1.1
    p.mg_nu1 = 2; p.mg_nu2 = 2; p.mg_nub = 8; p.mg_max_iter = 100
    p.hg_max_iter = 100; p.hg_nu1 = 2; p.hg_nu2 = 1; p.hg_nub = 8; p.hg_omega = 0.9
    p.mac_rel_eps = 1.0e-10; p.hg_rel_eps = -1.0; p.abort_on_max_iter = 1; p.hg_fmg = 1; p.mac_fmg = 1; p.hg_omega_pre1 = 1.45; p.hg_omega_pre2 = 0.7; p.hg_omega_fac1 = 1.6; p.hg_omega_fac2 = 0.9; p.hg_omega_fac3 = 0.65; p.mg_predict = 1
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError("vdn_params has no field %r" % k)
        setattr(p, k, v)
    return p


class Box(C.Structure):
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3)]


# every symbol include/varden_amd.h declares: (restype, argtypes)
_VP = C.c_void_p
_PI = C.POINTER(C.c_int)
_PD = C.POINTER(C.c_double)
_PVP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "vdn_params_default": (None, [C.POINTER(Params)]),
    "vdn_init": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int]),
    "vdn_finalize": (C.c_int, []),
    "vdn_last_error": (C.c_char_p, []),
    "vdn_build_flavour": (C.c_char_p, []),
    "vdn_last_stale_hip_error": (C.c_int, [C.c_int]),
    "vdn_set_stream": (C.c_int, [_VP]),
    "vdn_device_synchronize": (C.c_int, []),
    "vdn_get_params": (C.c_int, [C.POINTER(Params)]),
    "vdn_arena_stats": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vdn_set_extruded_2d": (C.c_int, [C.c_int]),
    "vdn_comm_get_unique_id": (C.c_int, [C.c_char_p]),
    "vdn_comm_init": (C.c_int, [C.c_char_p]),
    "vdn_comm_finalize": (C.c_int, []),
    "vdn_comm_nranks": (C.c_int, [C.POINTER(C.c_int)]),
    "vdn_comm_stats": (C.c_int, [C.POINTER(C.c_long), C.c_int]),
    "vdn_comm_transport": (C.c_char_p, []),
    "vdn_debug_switches": (C.c_char_p, []),
    "vdn_plan_describe": (C.c_int, [C.POINTER(Box), _PI, C.c_int, C.POINTER(Box), _PI, C.c_int, C.c_int, _PI, C.c_int,
                                    C.POINTER(C.c_long), C.c_int, _PI, _PI]),
    "vdn_box_candidates": (C.c_int, [C.c_int, C.POINTER(Box), _PI, _PI, C.c_int, _PI, C.c_int, _PI]),
    "vdn_layout_create": (C.c_int, [C.c_int, _PI, C.POINTER(Box), _PI, C.POINTER(Box), _PI, _PI, _PVP]),
    "vdn_layout_destroy": (C.c_int, [_VP]),
    "vdn_layout_nlevel": (C.c_int, [_VP]),
    "vdn_layout_nboxes": (C.c_int, [_VP, C.c_int]),
    "vdn_layout_nlocal": (C.c_int, [_VP, C.c_int]),
    "vdn_layout_global_index": (C.c_int, [_VP, C.c_int, C.c_int]),
    "vdn_layout_get_box": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(Box)]),
    "vdn_bc_tower_create": (C.c_int, [_VP, _PI, _PVP]),
    "vdn_bc_tower_destroy": (C.c_int, [_VP]),
    "vdn_bc_tower_phys": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vdn_bc_tower_adv": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vdn_bc_tower_ell": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vdn_multifab_create": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _PI, _PVP]),
    "vdn_multifab_destroy": (C.c_int, [_VP]),
    "vdn_multifab_nfabs": (C.c_int, [_VP]),
    "vdn_multifab_ncomp": (C.c_int, [_VP]),
    "vdn_multifab_nghost": (C.c_int, [_VP]),
    "vdn_multifab_get_box": (C.c_int, [_VP, C.c_int, C.POINTER(Box)]),
    "vdn_multifab_fab_size": (C.c_long, [_VP, C.c_int]),
    "vdn_multifab_dataptr": (C.c_int, [_VP, C.c_int, _PVP]),
    "vdn_multifab_copy_to_host": (C.c_int, [_VP, C.c_int, _VP]),
    "vdn_multifab_copy_from_host": (C.c_int, [_VP, C.c_int, _VP]),
    "vdn_multifab_setval": (C.c_int, [_VP, C.c_double, C.c_int, C.c_int, C.c_int]),
    "vdn_multifab_copy_c": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_int, C.c_int]),
    "vdn_multifab_norm_inf": (C.c_int, [_VP, C.c_int, C.c_int, _PD]),
    "vdn_multifab_min_max": (C.c_int, [_VP, C.c_int, _PD, _PD]),
    "vdn_multifab_fill_boundary": (C.c_int, [_VP]),
    "vdn_multifab_physbc": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _VP]),
    "vdn_advance_timestep": (C.c_int, [C.c_int, _VP, _PVP, _PVP, _PVP, _PVP, _PVP, _PVP, _PVP, _PVP, _VP,
                                       C.c_double, C.c_double, _PD, C.c_int, C.c_int]),
    "vdn_estdt": (C.c_int, [C.c_int, _VP, _VP, _VP, _VP, _PD, C.c_double, _PD]),
    "vdn_hgproject": (C.c_int, [C.c_int, _VP, _PVP, _PVP, _PVP, _PVP, _PVP, _PD, C.c_double, _VP, C.c_int]),
    "vdn_macproject": (C.c_int, [_VP, _PVP, _PVP, _PVP, _PD, _VP, C.c_int]),
    "vdn_ml_cc_restriction": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "vdn_ml_edge_restriction": (C.c_int, [_VP, _VP, C.c_int]),
    "vdn_multifab_fill_ghost_cells": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "vdn_create_umac_grown": (C.c_int, [_VP, _VP, C.c_int]),
    "vdn_ml_restrict_and_fill": (C.c_int, [C.c_int, _PVP, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "vdn_fillpatch": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "vdn_make_vorticity": (C.c_int, [_VP, C.c_int, _VP, C.POINTER(C.c_double), _VP]),
    "vdn_make_magvel": (C.c_int, [_VP, C.c_int, _VP]),
    "vdn_comm_allreduce_max": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "vdn_ml_nodal_prolongation": (C.c_int, [_VP, _VP]),
    "vdn_multifab_copy_layouts": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_int]),
    "vdn_tag_boxes": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_ubyte)]),
    "vdn_make_new_grids": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Box), _PI, C.POINTER(C.c_long)]),
    "vdn_last_step_timing": (C.c_int, [_PD]),
    "vdn_last_solver_stats": (C.c_int, [C.c_int, _PI, _PD, _PD]),
    "vdn_last_mac_level_form": (C.c_int, []),
    "vdn_k_slope": (C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP]),
    "vdn_k_velpred": (C.c_int, [_VP, _PVP, _VP, _PD, C.c_double, _VP]),
    "vdn_k_mkflux": (C.c_int, [_VP, _PVP, _PVP, _PVP, _VP, _VP, _PD, C.c_double, _VP, C.c_int, _PI]),
    "vdn_k_update": (C.c_int, [_VP, _PVP, _PVP, _PVP, _VP, _VP, _PD, C.c_double, C.c_int, _PI, _VP]),
    "vdn_k_mkvelforce": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_double, _VP]),
    "vdn_k_mkscalforce": (C.c_int, [_VP, _VP, _VP, C.c_double, _VP]),
    "vdn_k_make_at_halftime": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP]),
    "vdn_cc_solve": (C.c_int, [_VP, _VP, _PVP, _PD, _PI, C.c_double, C.c_double, C.c_int, _PI, _PD, _PD]),
    "vdn_cc_smooth": (C.c_int, [_VP, _VP, _PVP, _PD, _PI, C.c_int]),
    "vdn_nd_solve": (C.c_int, [_VP, _VP, _VP, _VP, _PD, _PI, C.c_double, C.c_double, C.c_int, _PI, _PD, _PD]),
    "vdn_bench_cc_smoother": (C.c_int, [_VP, _VP, _PVP, _VP, _PD, _PI, C.c_int, _PD, C.POINTER(C.c_long)]),
    "vdn_bench_cc_smoother_in_solve": (C.c_int, [_VP, _VP, _PVP, _VP, _PD, _PI, C.c_int, C.c_int, _PD, C.POINTER(C.c_long)]),
}

_lib = None


class VardenError(RuntimeError):
    pass


def load():
    """load libvarden_amd.so and type every entry point; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VardenError("HIP library %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().vdn_last_error()
        raise VardenError("varden_amd C-ABI error %d: %s" % (rc, (msg or b"?").decode()))
