"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/varden_amd.h declares, keeps its parameter defaults in sync with the Python mirror, and the
product path refuses to run (no CPU fallback) when no HIP device is visible."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "varden_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vdn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from varden_amd import capi
    lib = capi.load()                      # raises if the .so is missing or a typed symbol is absent
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "libvarden_amd.so does not export %s" % s
    assert set(syms) == set(capi.SIGNATURES), set(syms) ^ set(capi.SIGNATURES)
    # both builds of the library (csrc/Makefile): the product and the suite's, the same objects but for the switch table and the transport seam
    flav = {}
    for name in ("libvarden_amd.so", "libvarden_amd_testing.so"):
        L = C.CDLL(os.path.join(ROOT, "varden_amd", "csrc", name), mode=os.RTLD_LOCAL)
        for s in syms:
            assert hasattr(L, s), "%s does not export %s" % (name, s)
        L.vdn_build_flavour.restype = C.c_char_p
        flav[name] = L.vdn_build_flavour()
    assert flav == {"libvarden_amd.so": b"release", "libvarden_amd_testing.so": b"testing"}, flav
    assert lib.vdn_build_flavour() == capi.FLAVOUR.encode()


def test_release_build_reads_no_switch():
    """the shipped library's vdn_env never calls getenv: its switch table reports the release note, whatever the environment holds"""
    L = C.CDLL(os.path.join(ROOT, "varden_amd", "csrc", "libvarden_amd.so"), mode=os.RTLD_LOCAL)
    L.vdn_debug_switches.restype = C.c_char_p
    assert L.vdn_debug_switches().startswith(b"release build")


def test_param_defaults_in_sync_with_reference_parameters():
    from varden_amd import capi
    lib = capi.load()
    c = capi.Params()
    lib.vdn_params_default(C.byref(c))
    p = capi.default_params()
    for name, _ in capi.Params._fields_:
        a, b = getattr(c, name), getattr(p, name)
        if hasattr(a, "__len__"):
            assert [list(r) for r in a] == [list(r) for r in b], name
        else:
            assert a == b, name
    # reference src/_parameters: nscal 2, slope_order 4, use_minion F, cflfac 0.8, max_dt_growth 1.1, stencil_order 2
    assert (c.nscal, c.slope_order, c.use_minion, c.stencil_order) == (2, 4, 0, 2)
    assert (c.cflfac, c.max_dt_growth, c.visc_coef) == (0.8, 1.1, 0.0)
    assert c.mac_rel_eps == 1e-10        # macproject.f90:92


def test_no_cpu_fallback_and_error_convention():
    """without a GPU vdn_init must fail with a message (the Fortran shim turns that into `error stop`)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from varden_amd import capi
    lib = capi.load()
    p = capi.default_params()
    rc = lib.vdn_init(C.byref(p), 0, 1, 0)
    assert rc != 0
    assert lib.vdn_last_error()
    with pytest.raises(capi.VardenError):
        capi.check(rc)


def test_rejects_unimplemented_configurations():
    from varden_amd import capi
    lib = capi.load()
    p = capi.default_params(dm=1)
    assert lib.vdn_init(C.byref(p), 0, 1, 0) != 0 and b"dm must be 2 or 3" in lib.vdn_last_error()
    p = capi.default_params(diffusion_type=3)
    assert lib.vdn_init(C.byref(p), 0, 1, 0) != 0 and b"DIFFUSION" in lib.vdn_last_error()


def test_product_does_not_reference_the_oracle():
    """nothing under varden_amd/ may include, import or link anything under oracle/"""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "varden_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".f90", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                for ln in txt.splitlines():
                    if re.search(r"(#include|import|from|-l|-L).*(voracle|oracle/|vo\.h|libvoracle)", ln):
                        bad.append((fn, ln.strip()))
    assert not bad, bad


def test_namelist_parser_keeps_paths_and_quoted_commas():
    """a quoted value may hold '/' and ',' (plot_base_name = 'out/plt'); unquoted values still end at ',' '/' or the line"""
    from varden_amd.inputs import parse_namelist
    nl = parse_namelist("&PROBIN\n plot_base_name = 'out/plt'  ! comment\n check_base_name = \"a,b/chk\"\n n_cellx = 64, n_celly = 32\n"
                        " stop_time = 2.5d0\n use_minion = .true.\n u_bc(1,1) = 1.0\n fixed_grids = 'grids/g.txt'\n/\n")
    assert nl["plot_base_name"] == "out/plt" and nl["check_base_name"] == "a,b/chk" and nl["fixed_grids"] == "grids/g.txt"
    assert nl["n_cellx"] == 64 and nl["n_celly"] == 32 and nl["stop_time"] == 2.5 and nl["use_minion"] == 1 and nl["u_bc(1,1)"] == 1.0
