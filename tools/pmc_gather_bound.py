"""What bounds the hash gather sweep?  From the rocprofv3 counter summaries of profiles/run_pmc_tcp.sh (vector L1 / texture addresser) and
profiles/run_pmc_issue.sh (SQ instruction counters): per k_hash_fwd_pair variant, every candidate roof as a fraction of ITS rate over the
kernel's own cycles (SQ_BUSY_CYCLES / 32 shader engines):
    ta_busy            TA_TA_BUSY summed over the 256 addressers / 256 / cycles      (the addresser processes divergent 8-byte gathers at ~2 lanes per clock)
    tcp_busy           TCP_GATE_EN2 (vector-L1 core clock enabled) / 256 / cycles
    l1_tag_rate        TCP_TOTAL_CACHE_ACCESSES / 256 / cycles                        (1 tag lookup per clock and CU)
    valu_issue         4 * SQ_INSTS_VALU / (1024 SIMDs * cycles)                      (a wave64 VALU instruction occupies a SIMD for 4 cycles)
usage: python tools/pmc_gather_bound.py <tcp_per_kernel.csv> <issue_per_kernel.csv> <out.json>"""
import csv
import json
import sys

tcp = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[1]))}
iss = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
out = {}
for k, t in tcp.items():
    if "k_hash_fwd_pair" not in k or k not in iss:
        continue
    cyc = float(iss[k]["kernel_cycles"])
    f = lambda name: float(t["avg_" + name])  # noqa: E731
    out[k] = {"kernel_cycles": cyc,
              "ta_busy": round(f("TA_TA_BUSY_sum") / 256 / cyc, 4),
              "tcp_busy": round(f("TCP_GATE_EN2_sum") / 256 / cyc, 4),
              "l1_tag_rate": round(f("TCP_TOTAL_CACHE_ACCESSES_sum") / 256 / cyc, 4),
              "l1_miss_frac": round(f("TCP_TCC_READ_REQ_sum") / f("TCP_TOTAL_CACHE_ACCESSES_sum"), 4),
              "ta_cycles_per_wave_instruction": round(f("TA_TA_BUSY_sum") / f("TA_TOTAL_WAVEFRONTS_sum"), 1),
              "valu_issue": float(iss[k]["valu_issue_frac"]),
              "wait_over_wave_cycles": float(iss[k]["wait_over_wave_cycles"])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
