"""LimitsNormalizer (oracle; test infrastructure).  mpd/datasets/normalization.py:144-167."""
import torch


class LimitsNormalizer:
    def __init__(self, mins, maxs):
        self.mins = torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = torch.as_tensor(maxs, dtype=torch.float32)

    def normalize(self, x):  # :149-154
        x = (x - self.mins) / (self.maxs - self.mins)
        return 2 * x - 1

    def unnormalize(self, x, eps=1e-4):  # :156-167: ONE whole-tensor max/min decides whether everything is clipped
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clip(x, -1, 1)
        x = (x + 1) / 2.0
        return x * (self.maxs - self.mins) + self.mins
