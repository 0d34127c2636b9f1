"""The field normalisers of mpd/datasets/normalization.py (oracle; test infrastructure): Identity :111-116, GaussianNormalizer :119-141,
LimitsNormalizer :144-167, SafeLimitsNormalizer :170-184, FixedLimitsNormalizer :187-195.  The reference builds each from the flattened
field X [N, dim] (Normalizer.__init__ :90-93: per-dimension min / max over the rows); `LimitsNormalizer(mins, maxs)` below takes the limits
directly (what the planning path holds), `from_data` builds any of them the reference's way."""
import torch


class LimitsNormalizer:
    kind = "limits"

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


class Identity:
    kind = "identity"

    def __init__(self, mins=None, maxs=None):
        self.mins, self.maxs = mins, maxs

    def normalize(self, x):
        return x

    def unnormalize(self, x):
        return x


class GaussianNormalizer:
    kind = "gaussian"

    def __init__(self, means, stds, mins=None, maxs=None):
        self.means, self.stds = torch.as_tensor(means, dtype=torch.float32), torch.as_tensor(stds, dtype=torch.float32)
        self.mins, self.maxs = mins, maxs

    def normalize(self, x):  # :137-138
        return (x - self.means) / self.stds

    def unnormalize(self, x):  # :140-141
        return x * self.stds + self.means


def from_data(name, X, **kw):
    """`eval(normalizer)(X)` of DatasetNormalizer.__init__ (:14-22) for a flattened field X [N, dim]."""
    X = torch.as_tensor(X, dtype=torch.float32)
    mins, maxs = X.min(dim=0).values, X.max(dim=0).values   # :90-93
    if name == "Identity":
        return Identity(mins, maxs)
    if name == "GaussianNormalizer":
        return GaussianNormalizer(X.mean(dim=0), X.std(dim=0), mins, maxs)   # :126-129 (unbiased std)
    if name == "LimitsNormalizer":
        return LimitsNormalizer(mins, maxs)
    if name == "SafeLimitsNormalizer":
        # :175-184: the loop widens EVERY dimension by eps as soon as it meets a constant one (`self.mins -= eps` is the whole vector); after that
        # no dimension is constant any more (eps > 0), so it fires at most once
        eps = kw.get("eps", 1)
        if bool((mins == maxs).any()) and eps != 0:
            mins, maxs = mins - eps, maxs + eps
        return LimitsNormalizer(mins, maxs)
    if name == "FixedLimitsNormalizer":   # :192-195
        return LimitsNormalizer(torch.ones_like(mins) * kw.get("min", -1), torch.ones_like(maxs) * kw.get("max", 1))
    raise NameError(name)   # (what eval() of an unknown name raises)
