#!/usr/bin/env python3
"""Back-to-back timing of single conv layers (development tool; needs a GPU): python tools/time_layer.py B D layer-name-substring [env=val ...]
Prints, per matching layer and per value of MPDX_WS (0: per-layer kernel, 1: weight-stationary kernel where it applies), the time
per launch and the achieved fp32 TFLOP/s."""
import ctypes as C
import os
os.environ["MPDX_FUSED"] = "0"  # per-layer launch units: layer index == unit index
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch  # noqa: E402
from bench import build_model  # noqa: E402
from mpd_public_amd import _lib  # noqa: E402

B, D = int(sys.argv[1]), int(sys.argv[2])
filt = sys.argv[3:]
dm, sd = build_model(D, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, D, device="cuda")
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
cap = 128
ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); names = (C.c_char_p * cap)(); n = C.c_int()
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, B, ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)))
out = C.c_float()
for i in range(n.value - 1):
    nm = names[i].decode()
    if filt and not any(f in nm for f in filt):
        continue
    row = []
    for rep in range(2):
        for wsf in ("0", "1"):
            os.environ["MPDX_WS"] = wsf
            _lib.check(lib.mpdx_bench_layer(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), i, B, ws.data_ptr(), st, 50, 0, C.byref(out)))
            row.append(f"WS={wsf}: {out.value*1e3:8.1f} us {fl[i]/out.value/1e9:6.1f} TF/s")
    print(f"{nm:44s} {fl[i]/1e9:7.2f} GFLOP | " + " | ".join(row))
