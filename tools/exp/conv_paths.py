"""tests/test_convergence_gpu.py, scene 77: which of (precision, execution path) moves the trailing eikonal / normal terms?"""
import sys
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_convergence_gpu as T  # noqa: E402

make = T._teacher_scene()
tail = lambda h, k: float(h[k][-T.TAIL:].mean())  # noqa: E731
for prec, graph in (("fp32", True), ("bf16", False), ("bf16", True), ("fp32", False)):
    for scene in (77, 31):
        h, tr = T._fit(prec, graph, make(scene), seed0=9000)
        kinds = sorted(str(k) for k in tr._graphs) if graph else []
        print(f"{prec} graph={graph} scene {scene}: eikonal {tail(h, 'eikonal_loss'):.4f} normal_l1 {tail(h, 'normal_l1'):.4f} rgb {tail(h, 'rgb_loss'):.5f} "
              f"depth {tail(h, 'depth_loss'):.5f} loss {tail(h, 'loss'):.4f} {kinds}")
