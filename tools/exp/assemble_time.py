"""Where does hs_assemble on bf16 partial stacks spend its time?  (one job at a time, then the appearance backward's eleven together)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holoscene_amd.hashencoder.backend import _backend as be
dev, bf = "cuda", torch.bfloat16
S = 43
st = lambda r, c: torch.randn(S, r, c, device=dev).to(bf)
k_r2, k_r1, k_r0x, k_r0f, k_c1, k_c0 = st(32, 256), st(256, 256), st(256, 128), st(256, 256), st(256, 256), st(256, 128)
cs = [torch.randn(S, 256, device=dev) for _ in range(4)]
gb2 = torch.randn(3136, 4, device=dev)
enc = torch.randperm(128)[:81].to(torch.int32).to(dev)
fc = torch.randperm(128)[:32].to(torch.int32).to(dev)
gWr0 = torch.empty(256, 337, device=dev)
stk = lambda t, ld, col: (t, ld, col, t.shape[0], t[0].numel())
jobs = [((3, 256), [stk(k_r2, 256, 0)]), ((256, 256), [stk(k_r1, 256, 0)]), ((256, 81), [stk(k_r0x, 128, enc)], (gWr0, 0)),
        ((256, 256), [stk(k_r0f, 256, 0)], (gWr0, 81)), ((256, 256), [stk(k_c1, 256, 0)]), ((256, 32), [stk(k_c0, 128, fc)]),
        ((1, 256), [stk(cs[0], 0, 0)]), ((1, 256), [stk(cs[1], 0, 0)]), ((1, 256), [stk(cs[2], 0, 0)]), ((1, 256), [stk(cs[3], 0, 0)]),
        ((1, 3), [(gb2, 0, 0, 3136, 4)])]
def t(fn, n=50):
    """GPU time per call: n calls captured in ONE graph (no host time between the launches), replayed and bracketed by events."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for i, j in enumerate(jobs):
    print(i, j[0], "%.1f us" % t(lambda: be.assemble([j])))
print("all", "%.1f us" % t(lambda: be.assemble(jobs)))
print("all but the wide one", "%.1f us" % t(lambda: be.assemble(jobs[:-1])))
print("sum_slices", "%.1f us" % t(lambda: be.sum_slices([k_r2, k_r1, k_r0x, k_r0f, k_c1, k_c0] + cs)))
