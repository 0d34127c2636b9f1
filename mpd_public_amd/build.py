"""Builds libmpdx.so (hipcc, gfx950) in-tree.  `python -m mpd_public_amd.build` or __graft_entry__.build().

The library is one host translation unit (csrc/mpdx.hip) plus one per kernel family (csrc/k_*.hip, see csrc/host.hpp): the TUs are
compiled in parallel into build/obj/*.o and linked; an object is rebuilt only when its source, one of the headers it (transitively)
includes, the compile flags or this script changed - so an A/B edit of one kernel family costs that family's compile, not the whole
library's (round 3: one 107-KB TU, 99 s per rebuild).
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
def _extra() -> list:
    # dev builds: -DMPDX_DEV_HOOKS (tools/*_trace.py, ablate_layers.py), -DMPDX_LOOP_ABLATION (tools/ablate_loop.py), kernel A/B knobs
    return os.environ.get("MPDX_BUILD_DEFS", "").split()


# A build with extra definitions is a DIFFERENT library: it gets its own object directory and must name its own output
# (MPDX_BUILD_OUT=build_ab/libmpdx_<tag>.so), so that a dev / A-B build can neither clobber nor be mistaken for the shipped one.
_TAG = hashlib.sha256(" ".join(_extra()).encode()).hexdigest()[:8] if _extra() else ""
LIB = Path(os.environ["MPDX_BUILD_OUT"]).resolve() if os.environ.get("MPDX_BUILD_OUT") else PKG / "libmpdx.so"
OBJ = PKG.parent / "build" / ("obj" + ("_" + _TAG if _TAG else ""))
SOURCES = [CSRC / "mpdx.hip"] + sorted(CSRC.glob("k_*.hip"))
HEADERS = sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mpdx.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]

_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmpdx.so)")


def deps(src: Path) -> list:
    """src and every quoted header it includes, transitively."""
    seen, todo = {}, [src.resolve()]
    while todo:
        p = todo.pop()
        if p in seen or not p.exists():
            continue
        seen[p] = True
        for inc in _INC.findall(p.read_text(errors="replace")):
            todo.append((p.parent / inc).resolve())
    return sorted(seen)


def _stamp(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + _extra()).encode())
    h.update(Path(__file__).read_bytes())
    for p in deps(src):
        h.update(p.name.encode())   # (names, not absolute paths: the snapshot on the GPU box lives elsewhere)
        h.update(p.read_bytes())
    return h.hexdigest()


def _obj(src: Path) -> Path:
    return OBJ / (src.stem + ".o")


def stale(src: Path) -> bool:
    o, s = _obj(src), _obj(src).with_suffix(".stamp")
    return not (o.exists() and s.exists() and s.read_text() == _stamp(src))


def needs_build() -> bool:
    if not LIB.exists():
        return True
    if OBJ.exists() and all(_obj(s).exists() for s in SOURCES):
        t = LIB.stat().st_mtime
        return any(stale(s) or _obj(s).stat().st_mtime > t for s in SOURCES)
    # a shipped .so without its objects (the GPU box gets the .so of the snapshot): rebuild only if a source is newer
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def _compile(src: Path, verbose: bool) -> None:
    cmd = [hipcc()] + FLAGS + _extra() + ["-c", "-o", str(_obj(src)), str(src)]
    if verbose:
        print("[mpdx build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    _obj(src).with_suffix(".stamp").write_text(_stamp(src))


def build(force: bool = False, verbose: bool = True, only: list | None = None) -> Path:
    if _extra() and not os.environ.get("MPDX_BUILD_OUT"):
        raise RuntimeError("MPDX_BUILD_DEFS is set: name the output with MPDX_BUILD_OUT=build_ab/libmpdx_<tag>.so (a build with extra "
                           "definitions never replaces mpd_public_amd/libmpdx.so)")
    if not force and not only and not needs_build():   # `only` rebuilds the named TUs whatever their stamps say
        return LIB
    OBJ.mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    todo = [s for s in SOURCES if force or stale(s)]
    if only:   # dev: python -m mpd_public_amd.build k_fused  (rebuild these TUs whatever their stamps say)
        todo = sorted(set(todo) | {s for s in SOURCES if s.stem in only})
    jobs = int(os.environ.get("MPDX_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda s: _compile(s, verbose), todo))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(_obj(s)) for s in SOURCES]
    if verbose:
        print("[mpdx build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")]
    build(force="--force" in sys.argv, only=names or None)
    print(LIB)
