"""Would a streaming optimiser-style kernel (hs_adam_flat over one 48.8 MB table: 7 streams, ~60 us) overlap the latency-bound table scatter
(hs_hash_bwd_jac + reduction, ~85 us) if the two sat on parallel branches of the iteration graph?  Serial vs forked, both captured."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from holoscene_amd.hashencoder import HashEncoder, backend as B_   # noqa: E402

be = B_._backend
torch.manual_seed(1)
enc = HashEncoder(desired_resolution=2048).cuda()
R, N = 1024, 98
o = torch.rand(R, 1, 3, device="cuda") * 0.2 + 0.4
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device="cuda"), dim=-1)
z = torch.sort(torch.rand(R, N, 1, device="cuda") ** 3 * 0.5, 1)[0]
x = (o + z * d).reshape(-1, 3).clamp(0, 1).contiguous()
B, L, C = x.shape[0], 16, 2
S, H = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
g = torch.randn(L, B, C, device="cuda")
g[:, ::4] = 0
gj = torch.randn(L, B, 3 * C, device="cuda")
n = enc.embeddings.numel()
ge = torch.zeros(n, device="cuda")
p, gr, m, v = (torch.randn(n, device="cuda") * 0.1 for _ in range(4))
v.abs_()
st = B_.hsAdamState()
st.step = 0
st.group_end[0], st.group_end[1] = n, n
for i in range(3):
    st.lr0[i] = st.lr[i] = 1e-3
state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
be.adam_tick(state, 0.9, 0.99, 0.9999)
side = torch.cuda.Stream()


def scatter():
    be.bwd_jac(g, gj, x, enc.offsets, ge.view(-1, C), B, 3, C, L, S, H, ws=be.scatter_workspace(B, 3, C, L, "cuda"), level_major=True)


def stream_kernel():
    be.adam_flat(p, gr, m, v, 0, n, state, 0.9, 0.99, 1e-15, 1.0)


def body(mode, reps=20):
    for _ in range(reps):
        if mode == "scatter":
            scatter()
        elif mode == "stream":
            stream_kernel()
        elif mode == "serial":
            scatter(); stream_kernel()
        else:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                stream_kernel()
            scatter()
            cur.wait_stream(side)


for mode in ("scatter", "stream", "serial", "forked"):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(mode, 2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    be.scatter_workspaces_idle()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, capture_error_mode="thread_local"):
        body(mode)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 20)
    print(f"{mode:8s}: {sorted(ts)[2]:.1f} us per repetition")
