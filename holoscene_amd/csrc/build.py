"""Build libholoscene_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m holoscene_amd.csrc.build [--force]

The shared library lands next to the sources (git-ignored, but it travels to the GPU
box with the gpurun snapshot).  There is exactly one target: MI355X / gfx950.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libholoscene_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: the encoder must round exactly like the CPU oracle (no fused multiply-adds).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + [__file__]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False, variant=None, extra_flags=()):
    """variant: build a SECOND library `libholoscene_hip_<variant>.so` with extra compile flags (kernel A/B runs on one box:
    HOLOSCENE_LIB=<that path> selects it in hashencoder/backend.py); its objects live in their own directory."""
    if variant is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    lib = LIB if variant is None else os.path.join(HERE, f"libholoscene_hip_{variant}.so")
    objdir = HERE if variant is None else os.path.join(HERE, f"obj_{variant}")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", HERE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs]
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    # python -m holoscene_amd.csrc.build [--force] [--variant NAME -DFLAG ...]
    args = sys.argv[1:]
    var = args[args.index("--variant") + 1] if "--variant" in args else None
    print(build(force="--force" in args, verbose=True, variant=var, extra_flags=[a for a in args if a.startswith("-D")]))
