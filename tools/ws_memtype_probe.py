#!/usr/bin/env python3
"""Does the memory type of the activation workspace change the cost of a dependent-kernel boundary?  Times full U-Net
passes (B=100) with the workspace in (a) a torch tensor (hipMalloc, coarse-grained, L2-cached), (b) hipExtMallocWithFlags
fine-grained, (c) uncached (dev tool, needs a GPU)."""
import ctypes as C, sys, time
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
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
n_ws = lib.mpdx_unet_workspace_floats(hdl, B)
x = torch.randn(B, 64, 4, device="cuda"); eps = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
def run(ws_ptr, label):
    for _ in range(20):
        _lib.check(lib.mpdx_unet_forward(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, eps.data_ptr(), B, ws_ptr, st))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        _lib.check(lib.mpdx_unet_forward(hdl, packed.data_ptr(), tab.data_ptr(), 128, x.data_ptr(), 50, eps.data_ptr(), B, ws_ptr, st))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"{label:28s} {dt*1e6:8.1f} us per U-Net pass   eps checksum {float(eps.double().sum()):.6f}")
run(ws.data_ptr(), "torch tensor (coarse-grained)")
for flag, label in ((0x1, "fine-grained"), (0x3, "uncached")):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), n_ws * 4, flag)
    if rc != 0:
        print(label, "allocation failed", rc); continue
    run(p.value, label)
