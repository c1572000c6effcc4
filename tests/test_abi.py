"""CPU: the C-ABI library loads and exports every symbol include/holoscene_hip.h declares."""
import ctypes
import os
import re

from holoscene_amd.csrc import build as hip_build
from holoscene_amd.hashencoder import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "holoscene_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    hip_build.build()
    lib = ctypes.CDLL(backend.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/holoscene_hip.h but not exported"
    assert sorted(backend.dir_symbols()) == names


def test_library_identity():
    lib = backend.load_library()
    assert lib.hs_abi_version() >= 1
    assert lib.hs_target_arch() == b"gfx950"


def test_argument_errors_without_gpu():
    """Error paths return codes before any launch, so they are checkable on CPU."""
    lib = backend.load_library()
    lay = backend.hsHashLayout(2, 32, 6, 6, 0)
    one = ctypes.c_void_p(16)  # never dereferenced on these paths
    assert lib.hs_hash_fwd(one, one, one, one, 8, 4, 2, 16, ctypes.c_float(0.5), 16, None, ctypes.byref(lay), None) == -1  # D=4
    assert lib.hs_hash_fwd(one, one, one, one, 8, 3, 3, 16, ctypes.c_float(0.5), 16, None, ctypes.byref(lay), None) == -1  # C=3
    assert lib.hs_hash_fwd(None, one, one, one, 8, 3, 2, 16, ctypes.c_float(0.5), 16, None, ctypes.byref(lay), None) == -3
    assert lib.hs_hash_bwd2(one, one, one, 8, 3, 1, 16, ctypes.c_float(0.5), 16, one, one, one, one, ctypes.byref(lay), None) == -1  # C=1
    assert lib.hs_hash_fwd(one, one, one, one, 0, 3, 2, 16, ctypes.c_float(0.5), 16, None, ctypes.byref(lay), None) == 0  # empty batch


def test_cpu_tensor_is_rejected_loudly():
    import pytest
    import torch
    from holoscene_amd.hashencoder import HashEncoder
    enc = HashEncoder(num_levels=4, base_resolution=4, desired_resolution=32, log2_hashmap_size=10)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        enc(torch.zeros(4, 3))


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """Every structure the Python host mirrors with ctypes (hashencoder/backend.py) against the C header, compiled by gcc: same size, and
    every field at the same offset under the same name -- a field added to one side only (hsHashLayout grew three times this round)
    would otherwise show up as a silent argument shift on the GPU."""
    import ctypes
    import subprocess
    from holoscene_amd.hashencoder import backend as B
    names = [n for n in dir(B) if n.startswith("hs") and isinstance(getattr(B, n), type) and issubclass(getattr(B, n), ctypes.Structure)]
    assert {"hsHashLayout", "hsTableStep", "hsAdamState", "hsGate", "hsAsmJob", "hsWgradPairJob"} <= set(names)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "holoscene_hip.h"', "int main(void) {"]
    for n in names:
        lines.append(f'    printf("{n} size %zu\\n", sizeof({n}));')
        for f in getattr(B, n)._fields_:
            lines.append(f'    printf("{n} {f[0]} %zu\\n", offsetof({n}, {f[0]}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    want = {}
    for line in out:
        if line.strip():
            n, f, v = line.split()
            want[(n, f)] = int(v)
    for n in names:
        cls = getattr(B, n)
        assert ctypes.sizeof(cls) == want[(n, "size")], (n, ctypes.sizeof(cls), want[(n, "size")])
        for f in cls._fields_:
            assert getattr(cls, f[0]).offset == want[(n, f[0])], (n, f[0], getattr(cls, f[0]).offset, want[(n, f[0])])
