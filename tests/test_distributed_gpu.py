"""GPU: the data-parallel exchange with the REAL collectives and the real fused Adam kernel.

  * world size 1 on `nccl` (= RCCL): exactly the branch an 8-GPU run takes -- reduce_scatter_tensor into the staging slice,
    hs_adam_flat_shard on shard-local gradient / moment buffers, all_gather_into_tensor -- on the one GPU this box has;
  * world size 2 on `nccl` when two devices are visible (skipped otherwise);
  * world size 2 with both ranks sharing the GPU over `gloo` (RCCL refuses duplicate devices): the shard arithmetic of the HIP
    kernel (g_base / mv_base offsets) with two real shards.
Each case is compared with one process applying torch.optim.Adam to the mean of the ranks' gradients.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _TinyModel(torch.nn.Module):
    """The three optimiser groups of the Stage-1 model (tables | MLPs | beta) at toy sizes."""

    def __init__(self):
        super().__init__()
        self.grid = torch.nn.Parameter(torch.randn(50001, 2))
        self.net = torch.nn.Linear(37, 29)
        self.beta = torch.nn.Parameter(torch.tensor(0.1))
        me = self

        class NS:
            pass
        self.implicit_network, self.rendering_network, self.density = NS(), NS(), NS()
        self.implicit_network.grid_parameters = lambda: [me.grid]
        self.implicit_network.mlp_parameters = lambda: list(me.net.parameters())
        self.rendering_network.parameters = lambda: []
        self.density.parameters = lambda: [me.beta]


def _worker(rank, world, port, backend, q, zero1, steps, share_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0 if share_gpu else rank)
    torch.cuda.set_device(dev)
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    dist.init_process_group(backend, rank=rank, world_size=world)
    from holoscene_amd.training.distributed import exchange_and_step_flat
    from holoscene_amd.training.flat import FlatAdam
    torch.manual_seed(0)
    model = _TinyModel().to(dev)
    flat = FlatAdam(model, 5e-4, 20.0, 0.1, 1000, world_size=world, rank=rank, shard_moments=zero1)
    g = torch.Generator().manual_seed(50 + rank)
    hist = []
    used = torch.zeros(flat.padded, dtype=torch.bool)        # (the layout pads every table to a 16-byte boundary: pads carry zero gradients)
    for p_, off in zip(flat.params, flat.offsets):
        used[off:off + p_.numel()] = True
    for _ in range(steps):
        flat.zero_grad()
        local = torch.randn(flat.padded, generator=g) * used
        flat.flat_g.copy_(local.to(dev))
        hist.append(local.clone())
        exchange_and_step_flat(flat, world, zero1=zero1)
    torch.cuda.synchronize()
    full_m, full_v = flat.gather_moments()
    q.put((rank, [h.numpy() for h in hist], flat.flat_p.cpu().numpy().copy(), full_m.cpu().numpy().copy(), full_v.cpu().numpy().copy(), list(flat.offsets)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, backend, zero1, share_gpu, steps=3):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q, zero1, steps, share_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    flats = [r[2] for r in res]
    for f in flats[1:]:
        assert (f == flats[0]).all(), "replicas diverged"
    # single process: Adam on the mean of the ranks' gradients
    torch.manual_seed(0)
    ref = _TinyModel()
    opt = torch.optim.Adam([{"params": [ref.grid], "lr": 5e-4 * 20}, {"params": list(ref.net.parameters()), "lr": 5e-4},
                            {"params": [ref.beta], "lr": 5e-4}], betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.1 ** (1 / 1000))
    plist = [ref.grid] + list(ref.net.parameters()) + [ref.beta]
    offsets = res[0][5]          # flat index of each parameter's first element (optimiser order = plist order)
    for s in range(steps):
        mean = sum(torch.from_numpy(r[1][s]) for r in res) / world
        for p, off in zip(plist, offsets):
            p.grad = mean[off:off + p.numel()].view_as(p).clone()
        opt.step()
        sched.step()
    pick = lambda flat_arr: torch.cat([torch.from_numpy(flat_arr[off:off + p.numel()]) for p, off in zip(plist, offsets)])  # noqa: E731
    want_p = torch.cat([p.detach().reshape(-1) for p in plist])
    want_m = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in plist])
    want_v = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in plist])
    assert torch.allclose(pick(flats[0]), want_p, rtol=1e-5, atol=1e-7)
    for r in res:       # the gathered moments (what a checkpoint exports) are complete on every rank
        assert torch.allclose(pick(r[3]), want_m, rtol=1e-4, atol=1e-7)
        assert torch.allclose(pick(r[4]), want_v, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("zero1", [True, False])
def test_rccl_exchange_world_size_1(zero1):
    _run(1, "nccl", zero1, share_gpu=False)


@pytest.mark.parametrize("zero1", [True, False])
def test_rccl_exchange_two_gpus(zero1):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end multi-GPU runs take this branch)")
    _run(2, "nccl", zero1, share_gpu=False)


@pytest.mark.parametrize("zero1", [True, False])
def test_two_ranks_sharing_the_gpu_over_gloo(zero1):
    _run(2, "gloo", zero1, share_gpu=True)


# ---------------------------------------------------------------------------------------------------------------------
# The trainer's overlapped exchange: each hash table's segment exchanged on a side stream as soon as its last scatter has run
# (colour table under the trunk backward, SDF table under the trunk's weight-gradient GEMMs), all collectives captured INSIDE
# the whole-iteration graph.  One GPU: a one-rank RCCL group (data_parallel=True) takes
# exactly the code path of an N-rank run -- graph fork in the autograd thread, captured reduce_scatter_tensor /
# all_gather_into_tensor, segment-wise Adam -- and must reproduce the single-process trainer.
def _trainer_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

    def run(dp, exchange, steps=2):
        conf = stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision="bf16")
        tr = Stage1Trainer(conf, device=dev, optimizer="flat", graph=True, data_parallel=dp, exchange=exchange)
        benchmark_model_state(tr.model, 0.05)
        start = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
        scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=dev)
        fired = [0]
        if tr._overlap:
            orig = tr._exchange_early_segment

            def counted(s_):
                fired[0] += 1
                orig(s_)
            tr._exchange_early_segment = counted
        torch.manual_seed(77)
        for _ in range(steps):          # iteration 0 also runs the background-patch pass: two producers of the colour table's gradient
            tr.train_step_resident(scene)
        torch.cuda.synchronize()
        info = {"overlap": bool(tr._overlap), "segments": list(tr.flat.segments), "fired": fired[0], "graphs": len(tr._graphs),
                "first": [n for n, p in tr.model.named_parameters() if p is tr.flat.params[0]][0], "step": int(tr.flat.read_state().step)}
        return ({n: (p.detach() - start[n]).cpu().numpy() for n, p in tr.model.named_parameters()}, info)

    plain = run(False, None)
    over = run(True, "overlap")
    serial = run(True, "serial")
    # the fallback branch: this stack refusing a captured collective on a side stream -> warning, serial exchange after the graph
    import warnings
    probe = Stage1Trainer._probe_captured_collective
    Stage1Trainer._probe_captured_collective = lambda self: False
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        fallback = run(True, "overlap")
    Stage1Trainer._probe_captured_collective = probe
    fallback[1]["warned"] = any("exchanging serially" in str(w.message) for w in caught)
    q.put((plain, over, serial, fallback))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_overlapped_exchange_in_graph_equals_single_process():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_trainer_worker, args=(_free_port(), q))
    p.start()
    import queue as _queue
    got = None
    for _ in range(900):            # a worker that died must fail the test at once, not after the queue's whole timeout
        try:
            got = q.get(timeout=1.0)
            break
        except _queue.Empty:
            if not p.is_alive():
                break
    assert got is not None, f"the trainer worker exited without a result (exit code {p.exitcode})"
    (plain, pinfo), (over, oinfo), (serial, sinfo), (fallback, finfo) = got
    p.join(120)
    assert p.exitcode == 0
    assert not pinfo["overlap"] and len(pinfo["segments"]) == 1
    assert oinfo["overlap"] and len(oinfo["segments"]) == 3 and "color_encoding" in oinfo["first"]
    assert not sinfo["overlap"] and len(sinfo["segments"]) == 3
    assert not finfo["overlap"] and finfo["warned"] and len(finfo["segments"]) == 3 and finfo["step"] == 2, finfo
    # every captured variant (with / without the background pass) ran its body three times (two warm-ups + the capture), and in
    # every one of them both tables reported their gradients final -> both early segments went to the side stream each time
    assert oinfo["graphs"] == 2 and oinfo["fired"] == 2 * 3 * oinfo["graphs"], oinfo
    assert pinfo["step"] == oinfo["step"] == sinfo["step"] == 2
    for name, want in plain.items():
        scale = float(np.abs(want).max())
        for label, got in (("overlap", over[name]), ("serial", serial[name]), ("probe-failed fallback", fallback[name])):
            err = np.abs(got - want)
            # not bit-equal: the scatters' atomics sum in a different order from run to run, and Adam (eps = 1e-15) normalises every
            # element's step, so that noise grows with the horizon -- measured on the same trainer run twice: 3e-11 of the mean update
            # after one iteration, 1e-4 .. 3e-2 after three (the small colour-MLP matrices most), 0.2 after twelve
            # (tools/exp/dp_overlap_check.py).  Two iterations cover both graph variants; a segment that missed its update, or
            # stepped from a stale gradient, is an O(1) error at any horizon.
            assert float(np.mean(err)) <= 5e-2 * float(np.mean(np.abs(want))) + 1e-9, (label, name, float(np.mean(err)), scale)


# ---------------------------------------------------------------------------------------------------------------------
# The REAL trainer at world size 8 without an 8-GPU node: eight ranks share the one GPU over `gloo` (RCCL refuses duplicate devices, so the
# collectives cannot be captured: the exchange runs serially after the whole-iteration graph -- the form an RCCL run falls back to when its probe
# fails -- over the same three-segment ZeRO-1 layout).  Stock-size tables: 12 196 216 floats per grid (6 098 108 entries x 2; not a multiple of 8 ranks x 4
# floats), so shard bounds, staging slices, the sharded moments and the checkpoint export are exercised at the sizes an 8-GPU run has.
# Parity: after every step each rank's parameters equal Adam applied to the MEAN of the eight ranks' gradients -- the mean formed
# independently here by one all_reduce of the gradients each rank recorded before the exchange touched them.
def _trainer8_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from holoscene_amd.training import checkpoint as ck
    from holoscene_amd.training import distributed as dist_util
    from holoscene_amd.training import trainer as trainer_mod
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    conf = stock_conf(num_rays=128, S=32, d_out=4, num_levels=16, end_size=2048, logmap=19, beta=0.05, mlp_precision="bf16")
    tr = Stage1Trainer(conf, device=dev, world_size=world, rank=rank, optimizer="flat", graph=True, exchange="overlap")
    benchmark_model_state(tr.model, 0.05)
    dist_util.broadcast_parameters(tr.model)
    flat = tr.flat
    info = {"overlap": bool(tr._overlap), "segments": list(flat.segments), "shards": list(flat.shards), "padded": int(flat.padded),
            "moments": int(flat.flat_m.numel()), "table_step": bool(tr._table_step), "seg_mod": [(e - b) % world for b, e in flat.segments],
            "table_numel": [int(p.numel()) for p in flat.params[:flat.n_tables]], "n_params": sum(int(p.numel()) for p in flat.params)}
    # reference optimiser state, kept in full on every rank: torch ops, Adam's textbook form (training/holoscene_train.py:156-169: betas (0.9, 0.99), eps 1e-15)
    m_ref, v_ref = torch.zeros(flat.padded, device=dev), torch.zeros(flat.padded, device=dev)
    lr_of = torch.empty(flat.padded, device=dev)
    st0 = flat.read_state()
    ends = [int(st0.group_end[0]), int(st0.group_end[1]), flat.padded]
    lo = 0
    for g_, hi in enumerate(ends):
        lr_of[lo:hi] = float(st0.lr0[g_])
        lo = hi
    errs, rec = [], {}
    orig = dist_util.exchange_and_step_flat

    def recording(fl, ws, zero1=True, group=None, done=()):
        rec["p"], rec["g"] = fl.flat_p.clone(), fl.flat_g.clone()
        orig(fl, ws, zero1=zero1, group=group, done=done)
    trainer_mod.dist_util.exchange_and_step_flat = recording
    scene = SyntheticScene(128, 4, img_res=(64, 64), num_frames=3, ring=4, seed=1234 + rank, device=dev)
    steps = 3
    for t in range(1, steps + 1):
        tr.train_step_resident(scene)
        torch.cuda.synchronize()
        g_mean = rec["g"].clone()
        dist.all_reduce(g_mean)
        g_mean /= world
        m_ref.mul_(0.9).add_(g_mean, alpha=0.1)
        v_ref.mul_(0.99).addcmul_(g_mean, g_mean, value=0.01)
        gamma = 0.1 ** (1.0 / flat_decay_steps(tr))
        lr = lr_of * gamma ** (t - 1)
        want = rec["p"] - lr * (m_ref / (1 - 0.9 ** t)) / ((v_ref / (1 - 0.99 ** t)).sqrt() + 1e-15)
        upd = (want - rec["p"]).abs()
        err = (flat.flat_p - want).abs()
        mask = torch.zeros(flat.padded, dtype=torch.bool, device=dev)
        for p_, off in zip(flat.params, flat.offsets):
            mask[off:off + p_.numel()] = True
        errs.append((float(err[mask].max()), float(upd[mask].max()), float((err[mask] > 1e-3 * upd[mask] + 1e-9).float().mean()),
                     float((rec["g"][mask] != 0).float().mean())))
    full_m, full_v = flat.gather_moments()        # a collective: every rank
    mom = (float((full_m - m_ref).abs().max()), float(m_ref.abs().max()), float((full_v - v_ref).abs().max()), float(v_ref.abs().max()))
    ck.save_checkpoints(tr, tmp, 0, write=(rank == 0))     # ZeRO-1 export: every rank calls (moment gather + per-rank draw streams), rank 0 writes
    dist.barrier()
    saved_rng = torch.load(os.path.join(tmp, "ModelParameters", "latest.pth"))["hs_rng_state"]
    opt_sd = torch.load(os.path.join(tmp, "OptimizerParameters", "latest.pth"))["optimizer_state_dict"]
    n_state = sum(int(s_["exp_avg"].numel()) for s_ in opt_sd["state"].values())
    psum = float(flat.flat_p.double().sum())
    q.put((rank, info, errs, mom, tuple(saved_rng.shape), int(saved_rng[rank, 0]) == int(tr.model.rng_state(dev)[0]), n_state, psum, int(flat.read_state().step)))
    dist.barrier()
    dist.destroy_process_group()


def flat_decay_steps(tr):
    return float(tr.decay_steps)


def test_real_trainer_eight_ranks_sharing_the_gpu_stock_size_tables(tmp_path):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer8_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    res = []
    for _ in range(1500):
        try:
            res.append(q.get(timeout=1.0))
            if len(res) == world:
                break
        except _queue.Empty:
            if any(not p.is_alive() and p.exitcode not in (0, None) for p in procs):
                break
    assert len(res) == world, f"workers exited early: {[p.exitcode for p in procs]}"
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    info = res[0][1]
    # the layout of an 8-rank run at the stock grid: three segments (colour table | SDF table | MLPs + beta), no reduce-and-step, sharded moments
    assert not info["overlap"] and not info["table_step"] and len(info["segments"]) == 3
    assert info["table_numel"] == [12196216, 12196216] and all(m_ == 0 for m_ in info["seg_mod"])      # segments padded to multiples of the world size
    covered = 0
    for s_, (b, e) in enumerate(info["segments"]):
        n = (e - b) // world
        for r_ in range(world):
            assert tuple(res[r_][1]["shards"][s_]) == (b + r_ * n, b + (r_ + 1) * n)
        covered += e - b
    assert covered == info["padded"] and info["moments"] == info["padded"] // world
    for rank, _, errs, mom, rng_shape, rng_mine, n_state, psum, step in res:
        assert step == 3 and rng_shape == (world, 3) and rng_mine
        assert n_state == info["n_params"]         # the export holds FULL moments of every parameter although each rank stores an eighth
        for t, (emax, umax, frac_bad, nz) in enumerate(errs):
            print(f"PARITY world-8 trainer rank {rank} step {t + 1}: max |p - Adam(mean gradient)| {emax:.3e} (largest update {umax:.3e}), "
                  f"{frac_bad:.2e} of the elements beyond 1e-3 of their update; {nz:.3f} of the local gradient non-zero")
            # (a gradient whose eight contributions cancel to the last bits may change sign with the order of summation -- gloo's reduce against the
            #  all_reduce that formed the reference mean -- and Adam's first steps are sign-like: such an element is off by up to twice its update)
            assert emax <= 2.1 * umax and frac_bad < 1e-5, (rank, t, emax, umax, frac_bad)
        assert mom[0] <= 1e-5 * mom[1] + 1e-12 and mom[2] <= 1e-5 * mom[3] + 1e-15, mom
    assert len({r[7] for r in res}) == 1, "replicas diverged"
