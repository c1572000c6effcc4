"""Per-kernel FETCH_SIZE / WRITE_SIZE averages from the rocprofv3 counter CSVs of profiles/run_pmc_r02.sh -> CSV + pmc_traffic.json."""
import collections, csv, glob, json, os, re, subprocess, sys

prefix, out = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", name)[0]


def load(tag, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(f"{prefix}{tag}_{counter}/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"]) + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


res = {}
for tag in ("", "_calib"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, (n, v) in load(tag, c).items():
            res.setdefault(k, {"dispatches": n})[c + "_KiB_avg"] = v / n
rows = sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KiB_avg", 0) + kv[1].get("WRITE_SIZE_KiB_avg", 0)) * kv[1]["dispatches"])
with open(os.path.join(out, "pmc_per_kernel.csv"), "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg\n")
    for k, d in rows[:60]:
        f.write(f'"{k}",{d["dispatches"]},{d.get("FETCH_SIZE_KiB_avg", float("nan")):.1f},{d.get("WRITE_SIZE_KiB_avg", float("nan")):.1f}\n')
print(open(os.path.join(out, "pmc_per_kernel.csv")).read()[:4000])
