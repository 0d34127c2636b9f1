#!/usr/bin/env python3
"""Cycle stamps of the weight-stationary conv kernel (dev tool: needs a -DMPDX_DEV_HOOKS build and a GPU): workgroup 0, waves 0 / 1,
all eight waves: tile 8 top | k-loop issued | barrier passed | tile 9 top (kept in registers, written at kernel end)."""
import ctypes as C, os, sys
os.environ["MPDX_FUSED"] = "0"
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
dm, sd = build_model(14, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, 14, device="cuda")
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
cap = 128
ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); names = (C.c_char_p * cap)(); n = C.c_int()
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, B, ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)))
i = [k for k in range(n.value) if names[k].decode().startswith("mid_block1.blocks.1")][0]
stamps = (C.c_longlong * 32)()
_lib.check(lib.mpdx_layer_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), i, B, ws.data_ptr(), st, stamps))
lab = ["tile 8 top", "k-loop issued", "barrier passed", "tile 9 top"]
t0 = min(stamps[w * 4] for w in range(8))
for w in range(8):
    v = [stamps[w * 4 + k] - t0 for k in range(4)]
    duty = "duty" if w < 4 else ""
    print(f"wave {w} (SIMD {w % 4}): " + "  ".join(f"{lab[k]} {v[k]}" for k in range(4)) + f"   k-loop {v[1] - v[0]}  to barrier {v[2] - v[1]}  barrier -> next top {v[3] - v[2]}  {duty}")
