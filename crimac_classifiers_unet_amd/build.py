"""Build the gfx950 C-ABI library (libcrimac_unet_hip.so) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container;
the built .so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libcrimac_unet_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
SOURCES = ["conv3x3.hip", "conv3x3_glds.hip", "igemm.hip", "upconv.hip", "wgrad.hip", "elementwise.hip", "pack.hip", "tiling.hip", "augment.hip", "labels.hip", "meta.hip", "calib.hip"]
def _headers():
    """Every header a source may include: csrc/*.h plus the public C-ABI header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(PKG_DIR), "include", "crimac_unet_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
         "-Wno-unused-value"] + os.environ.get("CRIMAC_HIPCC_EXTRA", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile csrc/*.hip for gfx950 and link the shared library.  Returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
