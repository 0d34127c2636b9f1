#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), CPU only:
#   tools/resource_usage.sh k_guide [extra hipcc flags]
cd "$(dirname "$0")/../mpd_public_amd/csrc" || exit 1
TU=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -c $TU.hip -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c "
import re, sys, subprocess
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass', ln) or re.search(r':\d+:\d+: remark:\s+(.*?) \[-Rpass', ln) or re.search(r':\d+:\d+:\s+(.*?) \[-Rpass', ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(.*', '', n)[:110]
    print(f\"{n:110s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPR Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} sgpr {r.get('TotalSGPRs','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}\")
"
