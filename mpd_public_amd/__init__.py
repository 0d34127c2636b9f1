"""mpd_public_amd - MI355X (gfx950) native guided reverse-diffusion trajectory sampler.

Drop-in for the planning loop of jacarvalho/mpd-public (GaussianDiffusionModel / TemporalUnet / ddpm_sample_fn /
guide), running hand-written HIP kernels behind the C ABI of libmpdx.so (include/mpdx.h).
"""
from .temporal_unet import TemporalUnet, UNET_DIM_MULTS  # noqa: F401
from .diffusion_model import GaussianDiffusionModel  # noqa: F401
from .sample_functions import ddpm_sample_fn, guide_gradient_steps, apply_hard_conditioning, extract  # noqa: F401

from .guides import GuideManagerTrajectoriesWithVelocity  # noqa: F401
from .planning import CostCollision, CostGPTrajectory, CostComposite, PlanningTask, make_env, make_robot  # noqa: F401
from .datasets import (TrajectoryDataset, LimitsNormalizer, SafeLimitsNormalizer, FixedLimitsNormalizer, GaussianNormalizer, Identity,  # noqa: F401
                       make_normalizer)

__all__ = ["TemporalUnet", "UNET_DIM_MULTS", "GaussianDiffusionModel", "ddpm_sample_fn", "guide_gradient_steps",
           "apply_hard_conditioning", "extract", "GuideManagerTrajectoriesWithVelocity", "CostCollision", "CostGPTrajectory",
           "CostComposite", "PlanningTask", "TrajectoryDataset", "LimitsNormalizer", "SafeLimitsNormalizer", "FixedLimitsNormalizer", "GaussianNormalizer", "Identity",
           "make_normalizer", "make_env", "make_robot"]
