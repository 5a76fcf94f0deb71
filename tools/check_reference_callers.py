#!/usr/bin/env python
"""Compile the reference's OWN callers of the hot path against the product's Fortran surface -- syntax and semantics only, in the build container.

    python tools/check_reference_callers.py [--out profiles/r06_reference_callers_syntax.txt]

`flang -fsyntax-only -I varden_amd/fortran <file>` on files of /root/reference/src READ IN PLACE (nothing is copied; no object or module file of the
reference is written -- with -fsyntax-only flang emits none, and the working directory is a scratch directory): src/advance_timestep.f90's callers
(varden.f90, initialize.f90, regrid.f90, main.f90) and the modules between them and the boundary.  The module path holds the product's own modules under the
reference's names (varden_amd/fortran/varden_boxlib.f90, varden_boxlib_ext.f90), built first.  The report lists, per file: the `use`d modules that are absent,
the names a present module lacks, and every other semantic error (generic resolution, argument mismatches) -- i.e. exactly what a maintainer still has to
provide or change to drop the reference's driver onto the library.  Not a test: the reference tree is not on the GPU box, and the result is a list, not pass / fail.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
FDIR = os.path.join(ROOT, "varden_amd", "fortran")
# the callers of the boundary (advance_timestep, estdt, hgproject, the containers) and the driver above them; advance_timestep.f90 itself is the routine the
# library REPLACES -- it is listed to show which of its inner modules the boundary hides
# (estdt.f90, makevort.f90, tag_boxes.f90, multifab_physbc.f90 ... sit BEHIND the boundary: their loops over host `dataptr` arrays are what the HIP kernels replace)
FILES = ["main.f90", "varden.f90", "initialize.f90", "regrid.f90", "advance_timestep.f90"]


def flang():
    for c in ("amdflang", "/opt/rocm/lib/llvm/bin/flang", "flang"):
        p = subprocess.run(["sh", "-c", "command -v %s" % c], capture_output=True, text=True).stdout.strip()
        if p:
            return p
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_reference_callers_syntax.txt"))
    args = ap.parse_args()
    fc = flang()
    if not fc or not os.path.isdir(REF):
        sys.exit("needs flang and %s (the build container)" % REF)
    subprocess.check_call(["make", "-s", "-C", FDIR])
    lines = ["# flang -fsyntax-only of the reference's callers of the hot path against varden_amd/fortran (tools/check_reference_callers.py)",
             "# compiler: %s" % subprocess.run([fc, "--version"], capture_output=True, text=True).stdout.splitlines()[0],
             "# files are read in place from %s; nothing of the reference is copied or built" % REF, ""]
    tot_missing = collections.Counter()
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            src = os.path.join(REF, f)
            r = subprocess.run([fc, "-fsyntax-only", "-I", FDIR, src], capture_output=True, text=True, cwd=tmp)
            errs = [ln for ln in r.stderr.splitlines() if ": error:" in ln and not ln.startswith("error: Semantic errors")]
            missing_mod, missing_name, other = [], collections.defaultdict(list), []
            for e in errs:
                m = re.search(r":(\d+):\d+: error: Cannot parse module file for module '(\w+)'", e)
                if m:
                    missing_mod.append((m.group(2), int(m.group(1)))); continue
                m = re.search(r":(\d+):\d+: error: '(\w+)' not found in module '(\w+)'", e)
                if m:
                    missing_name[m.group(3)].append(m.group(2)); continue
                m = re.search(r"%s:(\d+):\d+: error: (.*)" % re.escape(src), e)
                other.append(("%s:%s" % (f, m.group(1)), m.group(2)) if m else ("", e))
            uses = len(re.findall(r"^\s*use\s+\w+", open(src).read(), flags=re.M | re.I))
            lines.append("== src/%s: %s (%d `use` statements, %d absent modules, %d absent names, %d other errors)"
                         % (f, "CLEAN: compiles against the product's modules unchanged" if r.returncode == 0 else "does not compile yet", uses,
                            len(missing_mod), sum(len(v) for v in missing_name.values()), len(other)))
            if missing_mod:
                lines.append("   absent modules: " + ", ".join("%s (:%d)" % mm for mm in missing_mod))
                for mm, _ in missing_mod:
                    tot_missing[mm] += 1
            for mod, names in sorted(missing_name.items()):
                lines.append("   absent in %s: %s" % (mod, ", ".join(sorted(set(names)))))
            # the errors that FOLLOW from an absent module (untyped names, unknown derived types) say nothing new: count them, print the rest
            follow = [o for o in other if re.search(r"No explicit type declared|Derived type '\w+' not found|is not an object of derived type|implicitly typed|must be a derived type", o[1])]
            rest = [o for o in other if o not in follow]
            if follow:
                names = sorted(set(re.findall(r"'(\w+)'", " ".join(o[1] for o in follow))))
                lines.append("   %d errors that follow from absences (names no present module declares, types of absent modules): %s" % (len(follow), ", ".join(names)))
            for where, msg in rest[:40]:
                lines.append("   %s: %s" % (where, msg))
            if len(rest) > 40:
                lines.append("   ... %d more" % (len(rest) - 40))
            lines.append("")
    lines.append("== absent modules over all files (module: files that use it)")
    for mod, cnt in sorted(tot_missing.items(), key=lambda kv: (-kv[1], kv[0])):
        lines.append("   %-28s %d" % (mod, cnt))
    text = "\n".join(lines) + "\n"
    with open(args.out, "w") as fh:
        fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
