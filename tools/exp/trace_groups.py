"""Median duration of consecutive groups of N launches of one kernel in a rocprofv3 kernel trace (microbenchmarks that time several cases of one kernel)."""
import csv, glob, sys
pat, n = sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i in range(0, len(rows), n):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i:i + n])
    print(i // n, round(d[len(d) // 2] / 1000, 1), "us  (min", round(d[0] / 1000, 1), ")")
