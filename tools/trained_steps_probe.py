"""dev probe: the `trained` leg of bench.py (generate -> train -> plan on the trained weights) at several training lengths: guided / unguided collision-free rates"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for s in [int(v) for v in sys.argv[1:]] or [20000, 25000, 30000]:
    r = bench.trained_leg(torch.device("cuda"), steps=s)
    keep = {k: r.get(k) for k in r if any(t in k for t in ("free", "train_s", "loss", "steps"))}
    print(s, json.dumps(keep)[:600], flush=True)
