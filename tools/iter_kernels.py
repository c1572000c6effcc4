#!/usr/bin/env python3
"""tools/iter_kernels.py <kernel_trace.csv> -- every kernel of the last full iteration in launch order (duration us, name, grid)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam_flat" in r["Kernel_Name"]]
a, b = ad[-2] + 1, ad[-1] + 1
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:78]
    print("%7.1f  %-78s g=%s" % (d, n, r.get("Grid_Size_X", r.get("Grid_Size"))))
