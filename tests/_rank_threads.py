"""Several ranks inside ONE process (test infrastructure of tests/test_multirank_gpu.py).

A GPU box admits at most six processes on its card, and BASELINE.json's configs[2] / configs[4] are EIGHT ranks (2 x 2 x 2: seven peers per
rank, edge- and corner-only peers in the nodal halos, an eight-rank all-gather of the agglomerated multigrid tail).  libvarden_amd.so keeps one
rank per loaded library (its context, caches and communicator are library globals), so a process that hosts several ranks loads one PRIVATE
COPY of the library per rank (a copy of the .so under another file name is another library to the dynamic loader: own globals, own streams;
loaded RTLD_LOCAL so that no symbol of one copy interposes the other's) and runs each rank's driver in its own thread -- ctypes releases
the interpreter lock during every library call, and the test transport (tests/fake_rccl) keeps its group state per thread.  Each rank
gets its own copy of the Python mirror too (`varden_amd` imported under an alias name: the mirror keeps per-library module state)."""
import ctypes as C
import importlib.util
import os
import shutil
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def private_package(rank, tmpdir):
    """the varden_amd mirror bound to a private copy of libvarden_amd.so; returns the package module (use pkg.boxlib, pkg.driver, ...)"""
    name = "varden_amd_rank%d" % rank
    pdir = os.path.join(ROOT, "varden_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pdir, "__init__.py"), submodule_search_locations=[pdir])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[name] = pkg
    spec.loader.exec_module(pkg)
    capi = importlib.import_module(name + ".capi")
    copy = os.path.join(tmpdir, "libvarden_amd_rank%d_%d.so" % (rank, os.getpid()))
    shutil.copyfile(capi.LIB_PATH, copy)
    lib = C.CDLL(copy, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for fn_name, (res, args) in capi.SIGNATURES.items():
        fn = getattr(lib, fn_name)
        fn.restype = res
        fn.argtypes = args
    capi._lib = lib                                    # what capi.load() would have set
    capi.LIB_PATH = copy
    for sub in ("boxlib", "advance", "driver", "plotfile"):
        importlib.import_module(name + "." + sub)
    return pkg


def run_ranks(ranks, body, tmpdir):
    """body(rank, pkg) for every rank of this process: directly for one rank (pkg = the plain varden_amd), in threads for several"""
    if len(ranks) == 1:
        import varden_amd
        from varden_amd import boxlib, driver, plotfile  # noqa: F401
        body(ranks[0], varden_amd)
        return
    pkgs = {r: private_package(r, tmpdir) for r in ranks}
    errs = []

    def run(r):
        try:
            body(r, pkgs[r])
        except BaseException as e:                     # noqa: B036 -- reported by the main thread
            import traceback
            traceback.print_exc()
            errs.append((r, e))
            os._exit(3)                                # the other ranks of this process would wait for this one forever

    ts = [threading.Thread(target=run, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0][1]
