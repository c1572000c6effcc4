"""Steady-state GPU time of the fused pixel draw + batch gather (hs_draw_gather) at the benchmark's shape: 50 launches in one graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holoscene_amd.training.synthetic import SyntheticScene
sc = SyntheticScene(1024, 32, num_frames=8, ring=64, device="cuda")
R = 1024
dst_in = {"uv": torch.zeros(1, R, 2, device="cuda"), "pose": torch.zeros(1, 4, 4, device="cuda"), "intrinsics": torch.zeros(1, 4, 4, device="cuda")}
dst_gt = {"rgb": torch.zeros(1, R, 3, device="cuda"), "depth": torch.zeros(1, R, 1, device="cuda"), "normal": torch.zeros(1, R, 3, device="cuda"),
          "mask": torch.zeros(1, R, 1, device="cuda"), "segs": torch.zeros(1, R, 1, device="cuda", dtype=sc.segs.dtype)}
for _ in range(20):
    sc.write_batch(dst_in, dst_gt)      # every frame's plan exists
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for _ in range(50):
            sc.write_batch(dst_in, dst_gt)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); g.replay(); b.record(); torch.cuda.synchronize()
print("hs_draw_gather: %.2f us per launch (back to back)" % (a.elapsed_time(b) / 50 * 1e3))
