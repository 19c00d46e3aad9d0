#!/usr/bin/env python3
"""Build the REFERENCE's own Cython hot path into oracle/_ref/ (test infrastructure only).

Compiles /root/reference/CRISPResso2/{CRISPResso2Align,CRISPRessoCOREResources}.pyx *where they
lie* (sources are never copied into this repo) with the reference's flags (`-w -Ofast`,
reference setup.py:22-35) into the private package `oracle/_ref/c2ref/`.  Outputs (generated C,
.so) only go under oracle/_ref/, which is git-ignored but travels to the GPU box.

Used for: (1) pinning oracle/c2_oracle.c against the real reference, (2) generating
tests/golden/*.json, (3) bench.py's `cpu_baseline` leg (kind="reference").
Never imported by the product path (crispresso2_amd/).
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("C2_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(HERE, "_ref")
PKG = os.path.join(OUT, "c2ref")
MODS = ["CRISPResso2Align", "CRISPRessoCOREResources"]


def built():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return all(os.path.exists(os.path.join(PKG, m + suffix)) for m in MODS)


def build(force=False):
    if built() and not force:
        return True
    src_dir = os.path.join(REF, "CRISPResso2")
    if not os.path.isdir(src_dir):
        return False  # GPU box: only the prebuilt .so files are used
    import numpy
    os.makedirs(PKG, exist_ok=True)
    open(os.path.join(PKG, "__init__.py"), "w").close()
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    inc = [sysconfig.get_paths()["include"], numpy.get_include()]
    for m in MODS:
        pyx = os.path.join(src_dir, m + ".pyx")
        c_out = os.path.join(OUT, m + ".c")
        subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_out])
        so = os.path.join(PKG, m + suffix)
        cmd = ["gcc", "-shared", "-fPIC", "-w", "-Ofast", c_out, "-o", so]
        for i in inc:
            cmd += ["-I", i]
        subprocess.check_call(cmd)
    # score matrices are data read at run time by read_matrix(); keep a copy beside the .so
    for f in ("EDNAFULL", "BLOSUM62"):
        shutil.copy(os.path.join(src_dir, f), os.path.join(OUT, f))
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built" if ok else "reference sources not present; nothing built")
