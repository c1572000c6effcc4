"""Median k_wgrad_pairs duration by caller in a bench trace: the launch that follows k_appear2_bwd (colour branch) and the one that follows
k_trunk_bwd / k_rr_bwd_value (trunk)."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = {}
for prev, r in zip(rows, rows[1:]):
    if "k_wgrad_pairs" in r["Kernel_Name"]:
        key = "colour" if "appear" in prev["Kernel_Name"] else "trunk"
        groups.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for k, d in sorted(groups.items()):
    d.sort()
    out.append(f"{k} {d[len(d) // 2] / 1000:.1f} us (n={len(d)})")
print("  ".join(out))
