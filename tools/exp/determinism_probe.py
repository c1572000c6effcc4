"""Is the bf16 whole-iteration graph deterministic where it should be?  Two trainers from one state on the same batches and draws: iteration 0's
losses must agree bit for bit (its forward has no atomics), later iterations may differ by the order of the table scatters' float atomics."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_convergence_gpu as T
T.STEPS = int(os.environ.get("STEPS", 12))
make = T._teacher_scene()
for prec, graph in (("bf16", True), ("fp32", False)):
    runs = [T._fit(prec, graph, make(77), seed0=9000)[0] for _ in range(3)]
    for k in ("loss", "eikonal_loss"):
        a, b, c = (r[k] for r in runs)
        print(prec, k, "step0 bitwise equal:", bool(torch.equal(a[:1], b[:1]) and torch.equal(a[:1], c[:1])),
              " max rel diff by step:", " ".join(f"{float(max(abs(a[i]-b[i]), abs(a[i]-c[i])) / abs(a[i])):.1e}" for i in range(T.STEPS)))
