#!/usr/bin/env python3
"""Per-layer timing + phase ablation of the conv-block kernels (development tool; needs a GPU).
usage: python tools/ablate_layers.py [B] [layer-name-substring ...]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
filt = sys.argv[2:]
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
lib = _lib.load()
hdl, packed, tab, ws = dm.model.engine(100, B)
x = torch.randn(B, 64, 4, device="cuda")
# one full pass so every workspace slot holds finite data
dm.model(x, torch.full((B,), 50, device="cuda", dtype=torch.long))
cap = 128
ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); names = (C.c_char_p * cap)(); n = C.c_int()
_s = torch.cuda.Stream(); torch.cuda.set_stream(_s)
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, B, ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)))
buf = C.create_string_buffer(64)
out = C.c_float()
print(f"{'layer':44s} {'tile':12s} {'MFLOP':>8s} | {'full':>7s} {'-stage':>7s} {'-mfma':>7s} {'-epi':>7s} {'only-launch':>11s}  TF/s")
tot = 0.0
for i in range(n.value - 1):
    nm = names[i].decode()
    if filt and not any(f in nm for f in filt):
        continue
    lib.mpdx_unet_layer_tile(hdl, i, B, buf, 64)
    res = []
    for dbg in (0, 1, 2, 4, 7, 16, 23):
        _lib.check(lib.mpdx_bench_layer(hdl, packed.data_ptr(), tab.data_ptr() , x.data_ptr(), i, B, ws.data_ptr(), st, 200, dbg, C.byref(out)))
        res.append(out.value * 1e3)
    tot += res[0]
    print(f"{nm:44s} {buf.value.decode():12s} {fl[i]/1e6:8.1f} | " + " ".join(f"{r:7.2f}" for r in res[:4]) + f" {res[4]:11.2f}  {fl[i]/res[0]/1e6:5.1f}   graph: full {res[5]:6.2f} empty {res[6]:5.2f}")
print(f"sum of back-to-back per-launch times: {tot:.1f} us")
