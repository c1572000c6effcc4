#!/bin/bash
# dev experiment: time k_trunk_fwd / k_trunk_bwd with parts compiled out (run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-BASE HS_EXP_NO_EPILOGUE HS_EXP_NO_WLOAD HS_EXP_NO_MMA HS_EXP_NO_STORE}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -D$v -I include -I holoscene_amd/csrc -shared holoscene_amd/csrc/sdf_mlp.hip -o /tmp/libexp_$v.so
  HS_EXP_LIB=/tmp/libexp_$v.so python - <<PY
import ctypes, os, sys, torch
lib = ctypes.CDLL(os.environ['HS_EXP_LIB'])
M = 4 * 104448
dev = 'cuda'
bf = torch.bfloat16
X = (torch.randn(M, 96, device=dev) * 0.3).to(bf)
w0 = (torch.randn(256, 96, device=dev) * 0.05).to(bf); w1 = (torch.randn(256, 256, device=dev) * 0.05).to(bf); w2 = (torch.randn(32, 256, device=dev) * 0.05).to(bf)
w2t = w2.t().contiguous(); w1t = w1.t().contiguous(); w0t = torch.zeros(256, 256, device=dev, dtype=bf); w0t[:96] = w0.t(); gfe = torch.empty(M // 4, 32, device=dev); gdy = torch.empty(16, M // 4, 6, device=dev)
b0 = torch.zeros(256, device=dev); b1 = torch.zeros(256, device=dev); b2 = torch.zeros(32, device=dev)
H0 = torch.empty(M, 256, device=dev, dtype=bf); H1 = torch.empty(M, 256, device=dev, dtype=bf); Y = torch.empty(M, 32, device=dev)
g = (torch.randn(M, 32, device=dev)).to(bf); gA1 = torch.empty_like(H0); gA0 = torch.empty_like(H0); gb1 = torch.zeros(256, device=dev); gb0 = torch.zeros(256, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def fwd():
    lib.hs_trunk_mlp_fwd(p(X), p(w0), p(b0), p(w1), p(b1), p(w2), p(b2), 32, p(H0), p(H1), p(Y), ctypes.c_int64(M), None, None, None, None, 0, 0, ctypes.c_float(0.0), None)
def bwd():
    lib.hs_trunk_mlp_bwd(p(g), 32, p(H1), p(H0), p(w2t), p(w1t), p(gA1), p(gA0), p(gb1), p(gb0), p(w0t), p(gfe), p(gdy), 16, 2, ctypes.c_float(0.5), ctypes.c_int64(M), None, None)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / 10 * 1e3, 1)
B = 131072
x = torch.rand(B, 3, device=dev) * 2 - 1; feat = torch.randn(B, 32, device=dev) * 1e-3; out = torch.empty(B, device=dev)
def sdf():
    lib.hs_sdf_mlp_fwd(p(x), p(feat), p(w0), p(b0), p(w1), p(b1), p(w2), p(b2), 32, -1, ctypes.c_uint64(0), p(out), None, ctypes.c_int64(B), None, 0, None)
print("$v fwd", t(fwd), "us  bwd", t(bwd), "us  sdf_mlp", t(sdf), "us")
PY
done
