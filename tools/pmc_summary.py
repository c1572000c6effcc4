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


# ---- pmc_traffic.json: HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts a wide read at half
# its bytes -- reproduced by the calibration stream below --, WRITE_SIZE at face value); bench.py reads `per_kernel_launch_bytes`
by_name = collections.defaultdict(lambda: {"n": 0, "fetch": 0.0, "write": 0.0})
for k, d in res.items():
    name = k.split(" grid=")[0]
    n = d["dispatches"]
    by_name[name]["n"] += n
    by_name[name]["fetch"] += d.get("FETCH_SIZE_KiB_avg", 0.0) * n
    by_name[name]["write"] += d.get("WRITE_SIZE_KiB_avg", 0.0) * n
per_kernel, launch_bytes = {}, {}
for name, a in by_name.items():
    if not (name.startswith("k_") or name.startswith("calib")):
        continue
    f, w = a["fetch"] / a["n"], a["write"] / a["n"]
    # gather kernels: a random 8-byte gather that misses the L2 is one 64-byte fabric request, tallied at face value (calib_gather8: 59 B per
    # gather) -- only WIDE reads are halved; the gather sweeps read nothing else of size
    rd = 1 if name.startswith("k_hash_fwd") or name.startswith("calib_gather") else 2
    per_kernel[name] = {"dispatches": a["n"], "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "read_correction": rd,
                        "read_bytes_corrected": int(f * 1024 * rd), "write_bytes": int(w * 1024)}
    launch_bytes[name] = int(f * 1024 * rd + w * 1024)
if "k_hash_fwd_pair" in launch_bytes:
    launch_bytes["k_hash_fwd"] = launch_bytes["k_hash_fwd_pair"]        # bench.py's row name for the gather sweeps (all launches averaged)
detail = {k: {"dispatches": d["dispatches"], "bytes": int(d.get("FETCH_SIZE_KiB_avg", 0.0) * 2048 + d.get("WRITE_SIZE_KiB_avg", 0.0) * 1024)}
          for k, d in res.items() if k.startswith("k_hash") or k.startswith("k_rr") or k.startswith("k_wgrad")}
json.dump({"commit": os.environ.get("HS_COMMIT", "see the commit that added this file"),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes with --kernel-trace only (profiles/run_pmc_r02.sh <tag>), over "
                     "`python bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0 --steps 12 --warmup 2` (the replayed "
                     "whole-iteration graph: the counters see its kernels one by one) and over "
                     "tools/exp/pmc_calib.hip; bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), averaged over the dispatches",
           "per_kernel_launch_bytes": launch_bytes, "per_kernel": per_kernel, "per_kernel_and_grid": detail}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("pmc_traffic.json:", {k: v for k, v in sorted(launch_bytes.items(), key=lambda kv: -kv[1])[:14]})
