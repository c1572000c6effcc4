"""Reads a rocprofv3 kernel trace csv; for the last complete iteration lists the launches around the early exchange."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]  # noqa: E731
adam = [i for i, r in enumerate(rows) if "k_adam_flat" in r["Kernel_Name"]]
last = adam[-1]
prev = [i for i in adam if i < last - 5][-1] if len([i for i in adam if i < last - 5]) else 0
start = max(i for i in range(last) if "k_adam_tick" in rows[i]["Kernel_Name"])
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:last + 3]:
    n = short(r["Kernel_Name"])
    if any(k in n for k in ("adam", "trunk_bwd", "appear_bwd", "hash_bwd", "bin_reduce", "copy", "Cijk", "wgrad", "ccl", "Memcpy", "fill", "reduce")):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {n}")
