#!/bin/bash
# dev experiment: time k_sdf_mlp with parts compiled out (run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT
for v in BASE HS_EXP_NO_EPILOGUE HS_EXP_NO_WLOAD HS_EXP_NO_MMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -D$v -I include -shared holoscene_amd/csrc/sdf_mlp.hip -o /tmp/libexp_$v.so
  HS_EXP_LIB=/tmp/libexp_$v.so python - <<PY
import ctypes, os, sys, torch
sys.path.insert(0, '.')
lib = ctypes.CDLL(os.environ['HS_EXP_LIB'])
B = 131072
dev = 'cuda'
x = torch.rand(B, 3, device=dev) * 2 - 1
feat = torch.randn(B, 32, device=dev) * 1e-3
w0 = (torch.randn(256, 96, device=dev) * 0.05).bfloat16(); w1 = (torch.randn(256, 256, device=dev) * 0.05).bfloat16(); w2 = (torch.randn(32, 256, device=dev) * 0.05).bfloat16()
b0 = torch.zeros(256, device=dev); b1 = torch.zeros(256, device=dev); b2 = torch.zeros(32, device=dev)
out = torch.empty(B, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    lib.hs_sdf_mlp_fwd(p(x), p(feat), p(w0), p(b0), p(w1), p(b1), p(w2), p(b2), 32, -1, ctypes.c_uint64(0), p(out), None, ctypes.c_int64(B), None)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print("$v", round(s.elapsed_time(e) / 20 * 1e3, 1), "us")
PY
done
