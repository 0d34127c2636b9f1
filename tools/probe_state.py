"""dev probe: does a leg of bench.py's default run change the batch-32 training iteration measured later in the same process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda")
def tr(tag, **k):
    r = bench.training_leg(steps=100, **k)
    print(f"{tag:40s} batch-32 iteration {r['ms_per_train_step']} ms", flush=True)
which = sys.argv[1]
if which == "a":
    tr("fresh, baseline=True", baseline=True)
    tr("second, baseline=False", baseline=False)
    tr("third, baseline=True", baseline=True)
elif which == "b":
    bench.trained_leg(dev); tr("after trained_leg, baseline=True", baseline=True); tr("again baseline=False", baseline=False)
elif which == "c":
    bench.guided_leg(dev); bench.planner_baseline_leg(); tr("after guided+planner, baseline=True", baseline=True)
