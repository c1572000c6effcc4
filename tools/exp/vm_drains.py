"""Per kernel of a .hip file: how often the ISA waits for the vector-memory counter to reach ZERO, next to its loads / stores / scratch
reloads / MFMAs.  The counter retires in issue order, so `s_waitcnt vmcnt(0)` in front of a load's use also waits for every store before it;
a kernel that stores activations and reloads spilled loop invariants (or loads inside its epilogue slices) drains its store queue each time."""
import re, subprocess, sys, collections
src = sys.argv[1]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Iholoscene_amd/csrc", "-S", "--cuda-device-only", src, "-o", "/tmp/vm_drains.s"],
               capture_output=True, text=True)
asm = open("/tmp/vm_drains.s").read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    c = collections.Counter()
    for l in body:
        l = l.strip()
        if re.match(r"s_waitcnt.*vmcnt\(0\)", l): c["drain"] += 1
        elif l.startswith("global_store") or l.startswith("buffer_store"): c["st"] += 1
        elif l.startswith("global_load_lds"): c["dma"] += 1
        elif l.startswith("global_load") or l.startswith("buffer_load"): c["ld"] += 1
        elif l.startswith("scratch_load"): c["reload"] += 1
        elif "v_mfma" in l: c["mfma"] += 1
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:28]
    print(f"{short:28s} drains {c['drain']:3d}  loads {c['ld']:3d}  stores {c['st']:3d}  dma {c['dma']:3d}  scratch reloads {c['reload']:3d}  mfma {c['mfma']:3d}")
