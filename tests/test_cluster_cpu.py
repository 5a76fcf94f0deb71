"""the host half of make_new_grids (varden_amd/csrc/cluster.h: Berger-Rigoutsos clustering + the merge of its boxes) without a GPU: tests/cpp/cluster_check.cpp holds
the one-pass merge against the start-again-from-the-first-pair rule it replaces (same boxes, same order -- the committed box lists of the fixtures depend on that order) and
checks disjointness, coverage, nesting and efficiency of the clustered boxes on random tag lattices (a 2-D lattice every fifth case)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clustering_and_merge_pass(tmp_path):
    exe = str(tmp_path / "cluster_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "varden_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "cluster_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
