"""dev probe: the ORDER of the device-side dispatches of one training iteration (kernel names from a rocprofv3 --kernel-trace csv): which launches sit
next to the runtime's own copy kernels (__amd_rocclr_copyBuffer: 16-17 per iteration in profiles/r05_train*_kernel_stats.csv)?
usage: python tools/train_trace_probe.py run <B> <D> [graph 0|1] [dim_mults option]     (the workload: 6 iterations through TrainStep.step)
       python tools/train_trace_probe.py show <trace.csv> (print the last iteration's dispatch sequence)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if sys.argv[1] == "run":
    import os
    os.environ["MPDX_TRAIN_GRAPH"] = sys.argv[4] if len(sys.argv) > 4 else "0"
    import bench
    print(bench.training_leg(steps=6, B=int(sys.argv[2]), D=int(sys.argv[3]), opt=int(sys.argv[5]) if len(sys.argv) > 5 else 1, baseline=False)["ms_per_train_step"])
else:
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    last = max(i for i, n in enumerate(names) if "adam_kernel" in n)
    prev = max(i for i, n in enumerate(names[:last]) if "adam_kernel" in n)
    t0 = int(rows[prev + 1]["Start_Timestamp"])
    for r in rows[prev + 1:last + 1]:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.2f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f}  {r['Kernel_Name'][:110]}")
