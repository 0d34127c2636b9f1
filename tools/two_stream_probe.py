#!/usr/bin/env python3
"""Probe: does splitting the batch over two HIP streams (independent kernel chains) beat one chain? (dev tool)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import synthetic as syn
D, T, n0 = 4, 100, 5
hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform", 0.6)).cuda(), 63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform", 0.6)).cuda()}
extra = lambda t: 0.5
def run(models, streams, Bs, reps=5):
    for _ in range(2):
        for m, s, B in zip(models, streams, Bs):
            with torch.cuda.stream(s):
                m.plan(hc, B, 64, n0, None, extra, return_chain=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for m, s, B in zip(models, streams, Bs):
            with torch.cuda.stream(s):
                m.plan(hc, B, 64, n0, None, extra, return_chain=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
m1, _ = build_model(D, (1, 2, 4, 8), T, "cuda")
print("1 stream  B=100      : %.2f ms" % run([m1], [torch.cuda.Stream()], [100]))
ms = [build_model(D, (1, 2, 4, 8), T, "cuda")[0] for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]
print("2 streams B=50+50    : %.2f ms" % run(ms[:2], ss[:2], [50, 50]))
print("4 streams B=25x4     : %.2f ms" % run(ms, ss, [25] * 4))
print("2 streams B=100+100  : %.2f ms (two full plans)" % run(ms[:2], ss[:2], [100, 100]))
print("1 stream  B=50       : %.2f ms" % run([m1], [ss[0]], [50]))
