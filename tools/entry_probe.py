"""dev probe: the inference entry (mpd_public_amd.inference.experiment, the reference's scripts/inference/inference.py) - wall time of a call and where the HOST
time goes; the timed region the entry reports itself (t_total) against the whole call."""
import sys, os, time, tempfile, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpd_public_amd.inference import experiment
for model_id in ("EnvDense2D-RobotPointMass", "EnvSpheres3D-RobotPanda"):
    d = tempfile.mkdtemp(prefix="mpdx_entry_")
    with contextlib.redirect_stdout(io.StringIO()):
        experiment(model_id=model_id, n_samples=100, results_dir=d, debug=False)   # first call: library load, allocations
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        r = experiment(model_id=model_id, n_samples=100, results_dir=d, debug=False)
    pr.disable(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tt = r.get("t_total") if isinstance(r, dict) else None
    print(f"{model_id}: whole call {wall * 1e3:.1f} ms; the entry's own t_total {tt}", flush=True)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print("\n".join(s.getvalue().splitlines()[6:30]))
