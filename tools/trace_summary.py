#!/usr/bin/env python3
"""tools/trace_summary.py <kernel_trace.csv> -- timeline summary of the LAST iteration in a rocprofv3 kernel trace."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] or 'k_adam' in r['Kernel_Name']]
ends = [ad[j] for j in range(len(ad)) if j == len(ad) - 1 or ad[j + 1] - ad[j] > 50]
a, b = ends[-2] + 1, ends[-1] + 1
it = rows[a:b]
def short(n):
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::', '', n)
    return n[:88]
t0 = int(it[0]['Start_Timestamp'])
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('kernels', len(it), 'span ms', (int(it[-1]['End_Timestamp']) - t0) / 1e6, 'busy ms', sum(dur(r) for r in it) / 1e6)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60
small_n = small_t = 0
for r in it:
    d = dur(r) / 1e3
    if d > thr:
        if small_n:
            print(f"           ... {small_n} kernels < {thr:.0f} us, busy {small_t/1e3:.3f} ms")
            small_n = small_t = 0
        print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  {d:8.1f} us  {short(r['Kernel_Name'])}")
    else:
        small_n += 1; small_t += d
if small_n: print(f"           ... {small_n} kernels < {thr:.0f} us, busy {small_t/1e3:.3f} ms")
agg = collections.Counter()
for r in it: agg[short(r['Kernel_Name'])[:60]] += dur(r)
print('--- top by total time')
for k, v in agg.most_common(12): print(f"{v/1e6:7.3f} ms  {k}")
print('--- kernels below the threshold, by name')
small = collections.Counter(); cnt = collections.Counter()
for r in it:
    if dur(r) / 1e3 <= thr: small[short(r['Kernel_Name'])[:110]] += dur(r); cnt[short(r['Kernel_Name'])[:110]] += 1
for k, v in small.most_common(40): print(f"{v/1e3:7.1f} us  x{cnt[k]:3d}  {k}")
