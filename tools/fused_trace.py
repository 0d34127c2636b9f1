#!/usr/bin/env python3
"""Per-phase cycle stamps of the fused level kernels, all 4 waves of workgroup 0 (dev tool, needs a GPU).
Per op five stamps: op entered (descriptor decoded) | k-loop issued | statistics exchanged (barrier A) | epilogue done | end-of-op barrier (B)."""
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
    stamps = (C.c_longlong * 1024)(); n = C.c_int(); nops = C.c_int()
    if lib.mpdx_fused_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), seg, B, ws.data_ptr(), st, stamps, 1024, C.byref(n), C.byref(nops)):
        break   # no such segment
    _lib.check(lib.mpdx_fused_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), seg, B, ws.data_ptr(), st, stamps, 1024, C.byref(n), C.byref(nops)))
    W = [[stamps[w * 128 + i] for i in range(128) if stamps[w * 128 + i]] for w in range(4)]
    t0 = min(w[0] for w in W if w)
    print(f"segment {seg}: {nops.value} ops; wave 0 total {W[0][-1] - W[0][0]} ticks; stamps relative to the first wave's entry, per wave")
    names = ["entry", "loads issued", "zeros", "prologue barrier"]
    for oi in range(nops.value):
        names += [f"op{oi} entered", f"op{oi} k-loop", f"op{oi} barrier A", f"op{oi} epilogue", f"op{oi} barrier B"]
    nst = max(len(w) for w in W)
    for i in range(nst):
        row = [(w[i] - t0) if i < len(w) else -1 for w in W]
        print(f"  {names[i] if i < len(names) else 'final':>20s}: " + " ".join(f"{v:7d}" for v in row))
