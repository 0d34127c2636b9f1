"""Randomised configuration fuzz of the training step on one GPU (dev tool): random (batch, state_dim, horizon, dim_mults option, loss, parameterisation,
timesteps) - loss and EVERY gradient of p_losses against float64 autograd of the oracle (2e-4 of the tensor's max, the tests' tolerance), or a loud refusal.
python tools/fuzz_train.py [n_cases] [seed]"""
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch   # noqa: E402
import mpd_public_amd as m   # noqa: E402
from mpd_public_amd.trainer import TrainStep   # noqa: E402
from helpers import synth_sd, t, DIM_MULTS   # noqa: E402
from oracle import train as otrain   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    B = rng.choice([1, 2, 3, 5, 9, 17, 32, 33, 47, 64, 70])
    D = rng.choice([2, 4, 6, 14, 24])
    H = rng.choice([24, 32, 40, 48, 64, 64, 96, 128])
    opt = rng.choice([0, 1])
    loss_type = rng.choice(["l2", "l2", "l1"])
    pe = rng.choice([True, True, False])
    sched, T = rng.choice([("exponential", 25), ("exponential", 100), ("cosine", 12)])
    with_hc = rng.choice([True, True, False])
    desc = f"B={B} D={D} H={H} opt={opt} {loss_type} predict_epsilon={pe} {sched} T={T} hard_conds={with_hc}"
    try:
        sd = synth_sd(D, opt)
        net = m.TemporalUnet(n_support_points=H, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
        net.load_state_dict(sd, strict=True)
        dm = m.GaussianDiffusionModel(model=net, variance_schedule=sched, n_diffusion_steps=T, predict_epsilon=pe, loss_type=loss_type).cuda()
        x0, noise = t(f"ft_x0/{case}", (B, H, D), "uniform", 0.8), t(f"ft_noise/{case}", (B, H, D))
        hc = {0: t(f"ft_hc0/{case}", (B, D), "uniform", 0.7), H - 1: t(f"ft_hc1/{case}", (B, D), "uniform", 0.7)} if with_hc else {}
        tt = torch.tensor([rng.randrange(T) for _ in range(B)])
        ts = TrainStep(dm)
        loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
        ref_loss, ref = otrain.loss_and_grads(sd, x0, tt, hc, noise, T, variance_schedule=sched, predict_epsilon=pe, loss_type=loss_type, dtype=torch.float64)
        worst, wname = 0.0, ""
        ok = abs(float(loss) - float(ref_loss)) < 5e-6 * max(1.0, abs(float(ref_loss)))
        tol = 2e-4 if loss_type == "l2" else 2e-3
        for name, p in dm.model.named_parameters():
            g, r = p.grad.detach().cpu().double(), ref[name]
            rel = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-7)
            if rel > worst:
                worst, wname = rel, name
            ok = ok and bool(torch.isfinite(g).all()) and rel <= tol
        if not ok:
            bad += 1
        print(f"{'ok' if ok else 'MISMATCH'} case {case}: {desc}: loss {float(loss):.6f} vs {float(ref_loss):.6f}, worst relative gradient error {worst:.2e} ({wname})")
    except Exception as e:
        print(f"refused case {case}: {desc}: {type(e).__name__}: {str(e)[:150]}")
print(f"{n_cases} cases, {bad} mismatches")
