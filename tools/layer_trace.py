#!/usr/bin/env python3
"""s_memtime phase stamps of conv_block_kernel for selected layers (dev tool, needs a GPU; per-layer path)."""
import ctypes as C, os, sys
os.environ["MPDX_FUSED"] = "0"; os.environ["MPDX_PAIR"] = "0"
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
filt = sys.argv[2:] or ["downs.3.1", "mid_block1", "downs.2.1", "ups.0.1"]
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, 4, device="cuda")
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
cap = 128
ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); names = (C.c_char_p * cap)(); n = C.c_int()
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, B, ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)))
lab = ["entry->staged(own)", "stage barrier", "mfma loop", "mfma barrier", "partials barrier", "epilogue"]
print("cycles per phase, first workgroup | last workgroup (start offset of last wg vs first)")
for i in range(n.value - 1):
    nm = names[i].decode()
    if not any(f in nm for f in filt):
        continue
    stamps = (C.c_longlong * 32)()
    for rep in range(2):
        _lib.check(lib.mpdx_layer_trace(hdl, packed.data_ptr(), tab.data_ptr(), x.data_ptr(), i, B, ws.data_ptr(), st, stamps))
    a = [stamps[k] for k in range(7)]; b = [stamps[16 + k] for k in range(7)]
    da = [a[k + 1] - a[k] for k in range(6)]; db = [b[k + 1] - b[k] for k in range(6)]
    print(f"{nm:42s} total {a[6]-a[0]:6d} | {b[6]-b[0]:6d}  (last wg starts +{b[0]-a[0]}, ends +{b[6]-a[0]})")
    print("      " + "  ".join(f"{l}: {x_}/{y_}" for l, x_, y_ in zip(lab, da, db)))
