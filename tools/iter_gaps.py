#!/usr/bin/env python3
"""tools/iter_gaps.py <kernel_trace.csv> [k] -- busy time vs span of the k-th graph-replayed iteration: how much of an iteration is idle
gaps between kernels, and which kernels (grouped by name) fill the rest."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam_flat" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) // 2
a, b = ad[k - 1] + 1, ad[k] + 1
it = rows[a:b]
span = (int(it[-1]["End_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / 1e3
gaps = [(int(it[i + 1]["Start_Timestamp"]) - int(it[i]["End_Timestamp"])) / 1e3 for i in range(len(it) - 1)]
print(f"iteration {k}: {len(it)} kernels, span {span:.1f} us, busy {busy:.1f} us, gaps {sum(g for g in gaps if g > 0):.1f} us "
      f"(mean {sum(g for g in gaps if g > 0) / max(1, len(gaps)):.2f} us, max {max(gaps):.1f} us), overlap {-sum(g for g in gaps if g < 0):.1f} us")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in it:
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.split(r"[<(]", n)[0][:60]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:8.1f} us  x{c:3d}  {n}")
