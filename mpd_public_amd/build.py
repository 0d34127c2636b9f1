"""Builds libmpdx.so (hipcc, gfx950) in-tree.  `python -m mpd_public_amd.build` or __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libmpdx.so"
SOURCES = [CSRC / "mpdx.hip"]
HEADERS = sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mpdx.h"]   # every header mpdx.hip includes


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmpdx.so)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on",
           "-Wall", "-Wno-unused-function", "-o", str(LIB)] + [str(s) for s in SOURCES]
    cmd += os.environ.get("MPDX_BUILD_DEFS", "").split()   # dev builds: -DMPDX_DEV_HOOKS (tools/*_trace.py, ablate_layers.py), -DMPDX_LOOP_ABLATION (tools/ablate_loop.py)
    if verbose:
        print("[mpdx build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
