"""GPU: does the bf16 whole-iteration-graph path TRAIN like the reference's precision?  

Single-iteration parity bounds the bf16 colour-branch gradients only to a few per cent (ReLU mask flips, tests/test_model_gpu.py::
test_fused_appearance_backward_on_fp32_relu_masks_is_tight); what matters for a training path is that those differences do not
accumulate.  A learnable synthetic scene -- every pixel of three 64 x 64 frames rendered by a TEACHER model (distinct objects, its own
weights) -- is fitted for 300 iterations from one initial state by (a) the benchmarked path, Stage1Trainer(graph=True, bf16), and (b) the fp32
path on the same batches, on two scenes.  Adam with the reference's eps = 1e-15 takes sign-like steps, so any two runs decorrelate
element by element within a dozen iterations; what a correct low-precision path must share with fp32 is the OUTCOME:
(a) must end at the objective of (b) on the same scene AND at its balance between the regularisers -- see the test body.  Asserted, not printed:
no non-finite loss anywhere, all runs learn (the rgb term falls by more than half), the trailing objectives agree.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
STEPS, TAIL = 300, 40


def _conf(precision):
    from holoscene_amd.training.trainer import stock_conf
    return stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision=precision, use_bg_reg=True,
                      learning_rate=1.0e-4)


def _teacher_scene():
    """SyntheticScene whose rgb / depth / normal images are renderings of a teacher network (eval mode, fp32)."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state
    teacher = Stage1Trainer(_conf("fp32"), device=DEV, optimizer="torch", seed=7)
    benchmark_model_state(teacher.model, 0.05, seed=7)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():       # give the teacher's objects their own shapes and colours
        net = teacher.model.implicit_network
        l2 = net.lin2
        l2.weight_v[:4] += (0.05 * torch.randn(4, l2.weight_v.shape[1], generator=g) * float(l2.weight_v[:4].abs().mean())).to(DEV)
        l2.bias[:4] += (0.15 * torch.randn(4, generator=g)).to(DEV)
        enc = net.color_encoding          # colours from the table; the geometry stays smooth (distinct objects through lin2 only)
        enc.embeddings.copy_(((torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * 2e-2).to(DEV))
    model = teacher.model.eval()

    def make(seed):
        sc = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=8, seed=seed, device=DEV)
        npix = sc.H * sc.W
        with torch.no_grad():
            for f in range(sc.F):
                for a in range(0, npix, 1024):
                    uv = sc.uv_all[a:a + 1024][None]
                    out = model({"uv": uv, "intrinsics": sc.intrinsics, "pose": sc.poses[f][None]}, torch.tensor([f]))
                    sc.rgb[f, a:a + 1024] = out["rgb_values"].reshape(-1, 3)
                    sc.depth[f, a:a + 1024] = out["depth_values"].reshape(-1, 1)
                    sc.normal[f, a:a + 1024] = torch.nn.functional.normalize(out["normal_map"].reshape(-1, 3), dim=-1)
        return sc
    return make


def _fit(precision, graph, scene, seed0=5000):
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state
    tr = Stage1Trainer(_conf(precision), device=DEV, optimizer="flat", graph=graph, seed=42)
    benchmark_model_state(tr.model, 0.05)
    hist = {"loss": [], "rgb_loss": [], "eikonal_loss": [], "depth_loss": [], "normal_l1": []}
    for i in range(STEPS):
        torch.manual_seed(seed0 + i)
        _, lo = tr.train_step(*scene.next_batch())
        for k in hist:
            hist[k].append(lo[k].detach().clone())
    torch.cuda.synchronize()
    return {k: torch.stack(v).float().cpu() for k, v in hist.items()}, tr


RUNS = 5        # runs per scene and precision: the statistics below are MEDIANS over them


def test_bf16_graph_training_tracks_fp32_training():
    """A run of either precision ends away from its pack now and then (tools/exp/conv_tail_spread.py, twelve runs each: scene 77 bf16 objectives
    2.30 .. 2.48 against fp32's 2.319 .. 2.326, scene 31 fp32 one run of twelve at 2.66 / normal-L1 1.19 against 2.53 .. 2.59 / 1.02 .. 1.05; one bf16
    run in eight at 2.89): float atomics and Adam's sign-like steps make every run its own trajectory.  One bf16 run against one fp32 run, as this
    test compared until round 5, therefore failed about one time in eight.  What a precision path must share with the reference's is the
    DISTRIBUTION of outcomes, so the comparison is between the medians of RUNS runs each -- a systematic shift (round 4's single-plane W2: every bf16
    run at an Eikonal term of 0.30 against 0.52) moves the median by as much as it moved every run."""
    make = _teacher_scene()
    runs = {}
    for scene, seed0 in ((31, 5000), (77, 9000)):
        bfs, fps = [], []
        for r in range(RUNS):
            bf, tr_bf = _fit("bf16", True, make(scene), seed0=seed0)
            assert ("full", False, False) in tr_bf._graphs, "the bf16 run must have gone through the whole-iteration graph"
            bfs.append(bf)
            fps.append(_fit("fp32", False, make(scene), seed0=seed0)[0])
        runs[scene] = (bfs, fps)
    tail = lambda h, k: float(h[k][-TAIL:].mean())  # noqa: E731
    med = lambda hs, k: float(torch.tensor([tail(h, k) for h in hs]).median())  # noqa: E731
    bad = []
    for scene, (bfs, fps) in runs.items():
        for name, hs in (("bf16", bfs), ("fp32", fps)):
            for h in hs:
                for k, v in h.items():
                    assert bool(torch.isfinite(v).all()), (scene, name, k)
                first, last = float(h["rgb_loss"][:10].mean()), tail(h, "rgb_loss")
                assert last < 0.5 * first, (scene, name, "the run does not learn", first, last)
            print(f"PARITY convergence scene {scene} {name}: trailing objective of the {RUNS} runs " + " ".join(f"{tail(h, 'loss'):.3f}" for h in hs)
                  + "; Eikonal " + " ".join(f"{tail(h, 'eikonal_loss'):.3f}" for h in hs))
        for k in ("loss", "rgb_loss", "eikonal_loss", "depth_loss", "normal_l1"):
            a, b, start = med(bfs, k), med(fps, k), float(torch.tensor([float(h[k][:10].mean()) for h in fps]).median())
            print(f"PARITY convergence scene {scene} {k}: median bf16 {a:.5f} fp32 {b:.5f} bf16 / fp32 {a / max(abs(b), 1e-12):.3f}, start {start:.5f}")
            # What is asserted, on the medians: the OBJECTIVE within 10 % of the fp32 runs' on the same scene (observed <= 3.2 %); the rgb term, which
            # falls to 6 % of its start (below bf16's resolution of the colours), within 4 % of that start; and the SPLIT between the regularisers: the
            # Eikonal and normal-L1 terms within 15 % of the fp32 runs' on both scenes.  Round 4 could not assert the last one: on scene 77 every
            # bf16 run settled at an Eikonal term of 0.30 / normal 0.95-1.0 against fp32's 0.51-0.54 / 0.78.  The cause was ONE rounding: the
            # last trunk layer's matrix as a single bf16 plane in the value product of the rendered samples (geometric initialisation leaves its rows at
            # 0.11 +- 1e-4, bf16's grid there is 4.9e-4; tools/exp/conv_hybrid.py); k_rr_fwd now carries it as two planes.
            if k == "loss" and abs(a - b) > 0.10 * abs(b):
                bad.append((scene, k, a, b))
            if k == "rgb_loss" and abs(a - b) > 0.04 * abs(start):
                bad.append((scene, k, a, b))
            if k in ("eikonal_loss", "normal_l1") and abs(a - b) > 0.15 * abs(b):
                bad.append((scene, k, a, b, "regulariser split"))
            if not (0.5 * abs(b) <= abs(a) <= 2.0 * abs(b)) and abs(a - b) > 0.04 * abs(start):
                bad.append((scene, k, a, b, "factor two"))
    assert not bad, bad
