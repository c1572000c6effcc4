#!/usr/bin/env python3
"""tools/iter_kernels.py <kernel_trace.csv> [k] -- every kernel of the k-th iteration (default: the last; negative counts from the end;
bench.py ends with eager instrumented iterations, its graph-replayed ones sit in the middle) in launch order (duration us, name, grid)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam_flat" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -1
k = k if k >= 0 else len(ad) + k
a, b = ad[k - 1] + 1, ad[k] + 1
gap = (int(rows[a]["Start_Timestamp"]) - int(rows[a - 1]["End_Timestamp"])) / 1e3
print("# idle before the first kernel of this iteration: %.1f us" % gap)
print("# iteration %d of %d, span %.1f us" % (k, len(ad), (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:78]
    print("%7.1f  %-78s g=%s" % (d, n, r.get("Grid_Size_X", r.get("Grid_Size"))))
