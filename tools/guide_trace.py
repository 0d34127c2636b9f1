#!/usr/bin/env python3
"""s_memtime phase stamps of the guide kernel, all 8 waves of workgroup 0 (dev tool, needs a GPU and a -DMPDX_DEV_HOOKS build):
   MPDX_LIB=build_ab/libmpdx_dev.so python tools/guide_trace.py [B]   (B >= 512: the dense Panda variant, two workgroups per CU)"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import mpd_public_amd as m
from mpd_public_amd import _lib
from helpers import product_guide, obstacle_hugging_trajs
lab = {"RobotPointMass": ["entry->staged", "collision slice", "wait all waves", "gather+clip", "GP prior", "apply"],
       "RobotPanda": ["entry->staged", "FK->LDS", "forces (sphere group)", "wait all waves", "gather+clip", "GP prior", "apply"]}
for env_id, robot_id in (("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")):
    ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    x = obstacle_hugging_trajs(ds, B, seed="trace", scale=0.95).cuda()
    pg = product_guide(ds).cuda()
    gp = pg.device_params(x.device)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    stamps = (C.c_longlong * 128)()
    for rep in range(2):
        _lib.check(lib.mpdx_guide_trace(C.byref(gp), x.data_ptr(), flag.data_ptr(), B, 64, ds.state_dim, st, stamps))
    print(robot_id, f"B={B}")
    for w in range(8):
        v = [stamps[w * 16 + k] for k in range(8) if stamps[w * 16 + k]]
        d = [b - a for a, b in zip(v, v[1:])]
        print(f"  wave {w}: " + "  ".join(f"{l}: {c}" for l, c in zip(lab[robot_id], d)) + f"   total {v[-1]-v[0]}")
        fs = [stamps[w * 16 + 8 + k] for k in range(4) if stamps[w * 16 + 8 + k]]
        if fs:
            t = [stamps[w * 16 + 2]] + fs
            print("          per field (self, objects, workspace, extra): " + "  ".join(str(b - a) for a, b in zip(t, t[1:])))
