#!/usr/bin/env python3
"""What bounds the MFMA loop of the per-layer conv kernel?  Back-to-back launch time of selected layers with the loop's
weight-ring refills (dbg 8) and/or its B-fragment LDS reads (dbg 32) removed (dev tool, needs a GPU; per-layer path)."""
import ctypes as C, os, sys
os.environ["MPDX_FUSED"] = "0"; os.environ["MPDX_PAIR"] = "0"
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
filt = sys.argv[2:] or ["mid_block1.blocks.0", "downs.2.1.blocks.0", "ups.0.1.blocks.0"]
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, 4, device="cuda")
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
cap = 128
ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); names = (C.c_char_p * cap)(); n = C.c_int()
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, B, ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)))
out = C.c_float()
print(f"{'layer':44s} {'full':>7s} {'-refill':>8s} {'-Bread':>7s} {'-both':>7s} {'-loop':>7s}   (us per back-to-back launch)")
for i in range(n.value - 1):
    nm = names[i].decode()
    if not any(f in nm for f in filt):
        continue
    res = []
    for dbg in (0, 8, 32, 40, 2):
        _lib.check(lib.mpdx_bench_layer(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), i, B, ws.data_ptr(), st, 200, dbg, C.byref(out)))
        res.append(out.value * 1e3)
    print(f"{nm:44s} " + " ".join(f"{r:7.2f}" for r in res))
