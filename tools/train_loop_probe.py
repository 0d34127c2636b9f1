"""Wall-clock of a whole training RUN (trainer.train through train.experiment: loader + step + EMA), per step, against the native step alone.
python tools/train_loop_probe.py [steps]"""
import json
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch   # noqa: E402
import bench   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
root = tempfile.mkdtemp(prefix="mpdx_loop_")
import contextlib, os
with contextlib.redirect_stdout(sys.stderr):
    logs, rec = bench.train_small_model(root, steps=steps)
print(json.dumps({"loader": "torch DataLoader" if os.environ.get("MPDX_TORCH_DATALOADER") == "1" else "BatchGatherLoader", "steps": steps,
                  "train_s": rec["train_s"], "ms_per_step_whole_run": round(1e3 * rec["train_s"] / steps, 3), "loss": rec["diffusion_loss_first_last"]}))
