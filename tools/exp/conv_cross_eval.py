"""Scene 77 of tests/test_convergence_gpu.py: the models trained in bf16 and in fp32, each EVALUATED (one forward + loss, no update, same batches and
torch-generator draws, eager) in both precisions: is the lower Eikonal term of the bf16 run a property of the trained model or of how bf16 measures it?"""
import sys
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_convergence_gpu as T  # noqa: E402

make = T._teacher_scene()
for scene in (77, 31):
    for prec in ("bf16", "fp32"):
        h, tr = T._fit(prec, False, make(scene), seed0=9000)
        tr.freeze_parameters = True
        res = {}
        for ev in ("bf16", "fp32"):
            tr.model.implicit_network.set_mlp_precision(ev)
            tr.model.rendering_network.set_mlp_precision(ev)
            sc = make(scene + 1000)
            acc = {"eikonal_loss": 0.0, "normal_l1": 0.0, "rgb_loss": 0.0, "loss": 0.0}
            for i in range(20):
                torch.manual_seed(777 + i)
                _, lo = tr.train_step(*sc.next_batch())
                for k in acc:
                    acc[k] += float(lo[k]) / 20
            res[ev] = acc
        for ev, a in res.items():
            print(f"scene {scene} trained {prec} evaluated {ev}: eikonal {a['eikonal_loss']:.4f} normal_l1 {a['normal_l1']:.4f} rgb {a['rgb_loss']:.5f} loss {a['loss']:.4f}")
