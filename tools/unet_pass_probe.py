import sys, time, torch
sys.path[:0] = ['.', 'tests']
from bench import build_model
B = int(sys.argv[1])
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
x = torch.randn(B, 64, 4, device="cuda"); t = torch.full((B,), 50, device="cuda", dtype=torch.long)
for _ in range(5): dm.model(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): dm.model(x, t)
torch.cuda.synchronize(); print(B, (time.perf_counter() - t0) / 50 * 1e6, "us per U-Net pass")
