"""Checkpoint layout / optimizer-state conversion (training/checkpoint.py) against torch.optim.Adam + ExponentialLR, the pair
the reference saves (training/holoscene_train.py:226-246).  CPU only: no kernel runs."""
import os

import torch

from holoscene_amd.training import checkpoint as ck
from holoscene_amd.training.trainer import Stage1Trainer, stock_conf


def _trainer(optimizer):
    return Stage1Trainer(stock_conf(num_rays=16, S=8, d_out=3, num_levels=4, end_size=32, logmap=8, base_size=4), device="cpu", optimizer=optimizer)


def _fake_steps(tr, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        for p in tr.model.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
        tr.optimizer.step()
        tr.scheduler.step()


def test_reference_layout_and_round_trip_through_the_flat_optimizer(tmp_path):
    ref = _trainer("torch")
    _fake_steps(ref, 3)
    ck.save_checkpoints(ref, str(tmp_path), 7)
    for sub in ("ModelParameters", "OptimizerParameters", "SchedulerParameters"):
        assert sorted(os.listdir(tmp_path / sub)) == ["7.pth", "latest.pth"]
    saved = torch.load(tmp_path / "OptimizerParameters" / "latest.pth")
    assert set(saved) == {"epoch", "optimizer_state_dict"} and [g["name"] for g in saved["optimizer_state_dict"]["param_groups"]] == \
        ["encoding", "net", "density"]

    flat = _trainer("flat")
    assert ck.load_checkpoints(flat, str(tmp_path), "latest") == 7
    for (n, a), (_, b) in zip(flat.model.named_parameters(), ref.model.named_parameters()):
        assert torch.equal(a, b), n
    st = flat.flat.read_state()
    assert st.step == 3 and flat.iter_step == 3
    assert abs(st.lr0[0] - 5e-4 * 20) < 1e-9 and abs(st.lr0[1] - 5e-4) < 1e-10
    for p, (m, v) in zip(flat.flat.params, flat.flat.moment_views()):
        q = dict(ref.model.named_parameters())[[n for n, x in flat.model.named_parameters() if x is p][0]]
        assert torch.equal(m, ref.optimizer.state[q]["exp_avg"]) and torch.equal(v, ref.optimizer.state[q]["exp_avg_sq"])

    # and back: what the flat optimiser writes is what torch.optim.Adam / ExponentialLR would have written
    opt_sd, sched_sd = ck.optimizer_state_dicts(flat)
    ref_opt, ref_sched = ref.optimizer.state_dict(), ref.scheduler.state_dict()
    assert sched_sd["last_epoch"] == ref_sched["last_epoch"] == 3 and abs(sched_sd["gamma"] - ref_sched["gamma"]) < 1e-15
    for ga, gb in zip(opt_sd["param_groups"], ref_opt["param_groups"]):
        assert ga["params"] == gb["params"] and ga["betas"] == gb["betas"] and ga["eps"] == gb["eps"] and ga["name"] == gb["name"]
        assert abs(ga["lr"] - gb["lr"]) <= 1e-12 * gb["lr"] and abs(ga["initial_lr"] - gb["initial_lr"]) <= 1e-12 * gb["initial_lr"]
    assert opt_sd["state"].keys() == ref_opt["state"].keys()
    for i in ref_opt["state"]:
        assert float(opt_sd["state"][i]["step"]) == float(ref_opt["state"][i]["step"]) == 3.0
        assert torch.equal(opt_sd["state"][i]["exp_avg"], ref_opt["state"][i]["exp_avg"])
        assert torch.equal(opt_sd["state"][i]["exp_avg_sq"], ref_opt["state"][i]["exp_avg_sq"])
    # a torch optimizer of the reference accepts it
    fresh = _trainer("torch")
    ck.load_optimizer_state(fresh, opt_sd, sched_sd)
    assert fresh.scheduler.last_epoch == 3


def test_fresh_flat_optimizer_saves_an_empty_adam_state(tmp_path):
    flat = _trainer("flat")
    ck.save_checkpoints(flat, str(tmp_path), 0)
    sd = torch.load(tmp_path / "OptimizerParameters" / "0.pth")["optimizer_state_dict"]
    assert sd["state"] == {} and abs(sd["param_groups"][0]["lr"] - 1e-2) < 1e-12
    other = _trainer("flat")
    ck.load_checkpoints(other, str(tmp_path), 0)
    assert other.flat.read_state().step == 0 and float(other.flat.flat_m.abs().max()) == 0.0


# ---- the device-resident draw stream in a checkpoint: one row per data-parallel rank
class _StreamModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(2))

    def rng_state(self, device):
        if getattr(self, "_rng_state", None) is None:
            self._rng_state = torch.tensor([int(torch.randint(0, 2 ** 62, (1,)).item()), 0, 0], dtype=torch.int64)
        return self._rng_state


class _StreamTrainer:
    def __init__(self, world_size=1, rank=0):
        self.model, self.world_size, self.rank, self.dp, self.device = _StreamModel(), world_size, rank, world_size > 1, "cpu"


def test_single_process_stream_round_trips_and_other_world_sizes_keep_their_own_streams():
    a = _StreamTrainer()
    a.model.rng_state("cpu")[1] = 41
    saved = ck._gather_rng_states(a)
    assert saved.shape == (1, 3) and int(saved[0, 1]) == 41
    b = _StreamTrainer()
    ck._restore_rng_state(b, saved)
    assert torch.equal(b.model.rng_state("cpu"), a.model.rng_state("cpu"))
    # the one-row state of a single-process (or pre-fix) checkpoint must NOT be handed to every rank of a two-rank run
    c = _StreamTrainer(world_size=2, rank=1)
    mine = c.model.rng_state("cpu").clone()
    import pytest
    with pytest.warns(UserWarning, match="draw stream"):
        ck._restore_rng_state(c, saved.view(3))          # (the old 1-D form)
    assert torch.equal(c.model.rng_state("cpu"), mine)


def _stream_worker(rank, world, port, path, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5 + 7919 * (rank + 1))       # trainer.py: every rank reseeds after the common initialisation
    tr = _StreamTrainer(world, rank)
    tr.model.rng_state("cpu")[1] = 100 + rank      # ranks have advanced their counters differently
    allr = ck._gather_rng_states(tr)               # what save_checkpoints stores (every rank calls it; rank 0 writes)
    if rank == 0:
        torch.save({"hs_rng_state": allr}, path)
    dist.barrier()
    fresh = _StreamTrainer(world, rank)
    fresh.model.rng_state("cpu")
    ck._restore_rng_state(fresh, torch.load(path)["hs_rng_state"])
    q.put((rank, tr.model.rng_state("cpu").tolist(), fresh.model.rng_state("cpu").tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_resume_their_own_draw_streams(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, str(tmp_path / "rng.pth"), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, before0, after0), (_, before1, after1) = got
    assert after0 == before0 and after1 == before1        # every rank resumes ITS stream ...
    assert before0[0] != before1[0] and after0 != after1   # ... and they differ (seed and counter)
