#!/usr/bin/env python3
"""CPU numeric study (no GPU): error of a split-bf16 emulation of fp32 GEMM vs plain fp32 accumulation, at the dominant layer's
shape (K = 256 channels x 5 taps = 1280, weights ~ U(-1/sqrt(K), 1/sqrt(K)), activations ~ post-Mish O(1)).
bf16 MFMA runs at 16x the fp32 MFMA rate on gfx950, so 3 products = 5.3x, 6 products = 2.7x faster than v_mfma_f32_16x16x4_f32."""
import numpy as np
rng = np.random.default_rng(0)
K, M, N = 1280, 64, 256
W = (rng.uniform(-1, 1, (M, K)) / np.sqrt(K)).astype(np.float32)
X = rng.standard_normal((K, N)).astype(np.float32) * 0.6

def bf16(x):   # round-to-nearest-even truncation of fp32 to bf16, returned as fp32
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)

def split(x, n):
    parts, rem = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(rem); parts.append(p); rem = (rem - p).astype(np.float32)
    return parts

def acc32(terms):   # fp32 accumulation over K in blocks of 4 (MFMA 16x16x4-like), products exact in fp32 for bf16 inputs
    out = np.zeros((M, N), np.float32)
    for A, B in terms:
        for k in range(0, K, 4):
            out = (out + (A[:, k:k+4].astype(np.float32) @ B[k:k+4, :].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return out

truth = W.astype(np.float64) @ X.astype(np.float64)
scale = np.abs(truth).mean()
res = {}
res["fp32 MFMA (k-blocks of 4)"] = acc32([(W, X)])
w, x = split(W, 3), split(X, 3)
res["bf16 x1 (plain bf16)"] = acc32([(w[0], x[0])])
res["bf16 x3 (hi*hi + hi*lo + lo*hi)"] = acc32([(w[0], x[0]), (w[0], x[1]), (w[1], x[0])])
res["bf16 x6 (3-way split, terms to 2^-24)"] = acc32([(w[0], x[0]), (w[0], x[1]), (w[1], x[0]), (w[0], x[2]), (w[2], x[0]), (w[1], x[1])])
print(f"mean |y| = {scale:.4f}")
for k, v in res.items():
    e = np.abs(v.astype(np.float64) - truth)
    print(f"{k:40s} max abs err {e.max():.3e}   rms {np.sqrt((e**2).mean()):.3e}   (rms / mean|y| = {np.sqrt((e**2).mean())/scale:.2e})")
