#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> a small JSON that carries the fingerprint of the kernel sources it was
measured on, so that bench.py can take a class's launch duration from the COMMITTED profile when it was measured on the sources it runs
(the in-situ HIP-event figure includes the event pair and the launch gaps inside a run of launches: 8 % above rocprofv3's kernel time in
round 4) - as tools/pmc_traffic.py does for the PMC traffic.

usage: kernel_stats_json.py <kernel_stats.csv> <batch B> <command the stats were taken on> > profiles/rNN_<cfg>_kernel_stats.json"""
import csv
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402  (csrc_fingerprint: the same function the reader uses)

rows = list(csv.DictReader(open(sys.argv[1])))
out = {"source": f"rocprofv3 --kernel-trace --stats -- {sys.argv[3] if len(sys.argv) > 3 else 'python bench.py ...'} ; MI355X", "batch": int(sys.argv[2]),
       "csrc_fingerprint": bench.csrc_fingerprint(), "csv": Path(sys.argv[1]).name,
       "kernels": {r["Name"][:200]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
                   for r in rows if "mpdx::" in r["Name"]}}
print(json.dumps(out, indent=1))
