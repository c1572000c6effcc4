bash profiles/run_rocprof.sh it --steps 30 --warmup 15 >/dev/null 2>&1
T=$(find /tmp/prof_it -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam_flat" in r["Kernel_Name"]]
iv = [(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in zip(ad, ad[1:])]
# pick the first long iteration among graph-replayed ones (index >= 6)
k = next(i for i in range(8, len(iv)) if iv[i] > 5000)
a, b = ad[k] + 1, ad[k + 1] + 1
print("# bg iteration", k, "interval", iv[k], "kernels", b - a)
prev_end = int(rows[a - 1]["End_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    print("%7.1f %6.1f  %s" % ((e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = e
PY
