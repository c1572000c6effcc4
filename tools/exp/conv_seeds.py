"""How much of the trailing eikonal / normal terms of tests/test_convergence_gpu.py is the particular draw sequence?  Same scene, same path, other seeds."""
import sys
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_convergence_gpu as T  # noqa: E402

make = T._teacher_scene()
tail = lambda h, k: float(h[k][-T.TAIL:].mean())  # noqa: E731


def fit_bf16(scene_seed, philox_seed):
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state
    tr = Stage1Trainer(T._conf("bf16"), device=T.DEV, optimizer="flat", graph=True, seed=42)
    benchmark_model_state(tr.model, 0.05)
    tr.model.rng_state(T.DEV)[0] = philox_seed
    sc = make(scene_seed)
    hist = {"eikonal_loss": [], "normal_l1": [], "rgb_loss": [], "loss": []}
    for i in range(T.STEPS):
        _, lo = tr.train_step(*sc.next_batch())
        for k in hist:
            hist[k].append(lo[k].detach().clone())
    return {k: torch.stack(v).float().cpu() for k, v in hist.items()}


for scene in (31, 77):
    for ps in (1, 2, 3):
        h = fit_bf16(scene, 1000 + ps)
        print(f"bf16 scene {scene} philox seed {1000 + ps}: eikonal {tail(h, 'eikonal_loss'):.4f} normal_l1 {tail(h, 'normal_l1'):.4f} rgb {tail(h, 'rgb_loss'):.5f} loss {tail(h, 'loss'):.4f}")
    for s0 in (5000, 9000, 13000):
        h, _ = T._fit("fp32", False, make(scene), seed0=s0)
        print(f"fp32 scene {scene} seed0 {s0}: eikonal {tail(h, 'eikonal_loss'):.4f} normal_l1 {tail(h, 'normal_l1'):.4f} rgb {tail(h, 'rgb_loss'):.5f} loss {tail(h, 'loss'):.4f}")
