#!/usr/bin/env python3
"""Per-phase cycle stamps of the fused level kernels (dev tool, needs a GPU)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, 4, device="cuda")
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
st = torch.cuda.current_stream().cuda_stream
for seg in range(4):
    stamps = (C.c_longlong * 256)(); n = C.c_int(); nops = C.c_int()
    if lib.mpdx_fused_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), seg, B, ws.data_ptr(), st, stamps, 256, C.byref(n), C.byref(nops)):
        break   # no such segment
    _lib.check(lib.mpdx_fused_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), seg, B, ws.data_ptr(), st, stamps, 256, C.byref(n), C.byref(nops)))
    v = [stamps[i] for i in range(256) if stamps[i]]
    d = [b - a for a, b in zip(v, v[1:])]
    print(f"segment {seg}: {nops.value} ops, total {v[-1]-v[0]} cycles; prologue: ring + input loads issued {d[0]}, parameter loads issued {d[1]}, "
          f"halo zeros {d[2]}, loads landed + barrier {d[3]}")
    k = 4
    for oi in range(nops.value):
        if k + 3 < len(d) + 1:
            print(f"   op{oi}: k-loop {d[k]:6d}  stats+barrier {d[k+1]:6d}  epilogue {d[k+2]:6d}  barrier {d[k+3] if k+3 < len(d) else -1:6d}")
            k += 4
    print(f"   remaining stamps after the last printed op: {d[k:]}")
