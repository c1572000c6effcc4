#!/usr/bin/env python3
"""tools/fixed_vs_per_ray.py -- per-kernel cost of one iteration as  fixed + slope x rays,  from tools/exp/kstat_args.sh runs at several --rays
(the text that script prints, concatenated with '== --rays N' headers):   python tools/fixed_vs_per_ray.py gpurun_out/r06/kstat_rays.txt"""
import re
import sys
from collections import defaultdict

rows = defaultdict(dict)      # kernel -> rays -> us per iteration
rays = None
for line in open(sys.argv[1]):
    m = re.match(r"== --rays (\d+)", line)
    if m:
        rays = int(m.group(1))
        continue
    m = re.match(r"\s*([\d.]+) us/iter\s+x\s*([\d.]+)\s+avg\s+([\d.]+) us\s+(.*)", line)
    if m and rays:
        name = re.sub(r"\(.*", "", m.group(4)).strip()
        if "spin_kernel" in name:
            continue
        rows[name][rays] = rows[name].get(rays, 0.0) + float(m.group(1))
pts = sorted({r for v in rows.values() for r in v})
lo, hi = pts[0], pts[-1]
print(f"# us per iteration (all launches of a kernel summed; mean over regular and background-patch iterations) at --rays {pts};")
print(f"# fixed / per-ray: the line through the {lo}- and {hi}-ray points, t = fixed + slope x rays")
print(f"{'kernel':58s}" + "".join(f"{p:>9d}" for p in pts) + f"{'fixed us':>10s}{'us/kray':>9s}")
tot = defaultdict(float)
fixed_sum = slope_sum = 0.0
for name, v in sorted(rows.items(), key=lambda kv: -kv[1].get(1024, 0.0)):
    if lo not in v or hi not in v:
        continue
    slope = (v[hi] - v[lo]) / (hi - lo)
    fixed = v[lo] - slope * lo
    fixed_sum += fixed
    slope_sum += slope
    for p in pts:
        tot[p] += v.get(p, 0.0)
    print(f"{name[:58]:58s}" + "".join(f"{v.get(p, float('nan')):9.1f}" for p in pts) + f"{fixed:10.1f}{slope * 1000:9.1f}")
print(f"{'SUM of kernel time':58s}" + "".join(f"{tot[p]:9.1f}" for p in pts) + f"{fixed_sum:10.1f}{slope_sum * 1000:9.1f}")
