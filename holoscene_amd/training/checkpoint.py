"""Checkpoints in the reference's on-disk layout (training/holoscene_train.py:226-246 save, :174-197 load), so that a run
can be resumed by either code base:

    <checkpoints>/ModelParameters/{<epoch>,latest}.pth      {"epoch", "model_state_dict"}
    <checkpoints>/OptimizerParameters/{<epoch>,latest}.pth  {"epoch", "optimizer_state_dict"}   torch.optim.Adam format, 3 groups
    <checkpoints>/SchedulerParameters/{<epoch>,latest}.pth  {"epoch", "scheduler_state_dict"}   ExponentialLR format

The flat fused optimiser (training/flat.py) keeps its moments in two flat buffers and its step / learning rates in a
device struct; here they are converted to and from the per-parameter ``torch.optim.Adam`` state the reference stores.
Correspondence after n updates:  Adam state["step"] = n for every parameter;  param_group["lr"] = initial_lr * gamma^n
(the scheduler has been stepped n times, :428);  scheduler.last_epoch = n;  hsAdamState.step = n, lr0 = initial_lr.
"""
import os

import torch

from .optim import build_optimizer, build_scheduler

MODEL_DIR, OPTIMIZER_DIR, SCHEDULER_DIR = "ModelParameters", "OptimizerParameters", "SchedulerParameters"


def _torch_pair(trainer):
    opt = build_optimizer(trainer.model, trainer.lr, trainer.lr_factor)
    return opt, build_scheduler(opt, trainer.decay_rate, trainer.decay_steps)


def optimizer_state_dicts(trainer):
    """(optimizer_state_dict, scheduler_state_dict) in torch.optim.Adam / ExponentialLR format."""
    if trainer.flat is None:
        return trainer.optimizer.state_dict(), trainer.scheduler.state_dict()
    flat = trainer.flat
    opt, sched = _torch_pair(trainer)
    n = int(flat.read_state().step)
    # ZeRO-1: every rank holds real moments for its own slice only -> collect the slices first (a collective: every rank of the
    # job must call this, whichever rank writes the files)
    full = flat.gather_moments() if getattr(trainer, "zero1", False) else None
    if n > 0:
        for p, (m, v) in zip(flat.params, flat.moment_views(full)):
            opt.state[p] = {"step": torch.tensor(float(n)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
    for grp in opt.param_groups:
        grp["lr"] = grp["initial_lr"] * flat.gamma ** n
    sched.last_epoch = n
    sched._step_count = n + 1
    sched._last_lr = [grp["lr"] for grp in opt.param_groups]
    return opt.state_dict(), sched.state_dict()


def load_optimizer_state(trainer, optimizer_state_dict, scheduler_state_dict):
    if trainer.flat is None:
        trainer.optimizer.load_state_dict(optimizer_state_dict)
        trainer.scheduler.load_state_dict(scheduler_state_dict)
        return int(scheduler_state_dict.get("last_epoch", 0))
    flat = trainer.flat
    opt, _ = _torch_pair(trainer)
    opt.load_state_dict(optimizer_state_dict)       # maps the saved per-index state onto this model's parameters, checks group sizes
    gamma = float(scheduler_state_dict["gamma"])
    if abs(gamma - flat.gamma) > 1e-12 * abs(flat.gamma):
        raise ValueError(f"checkpoint decays the learning rate by {gamma!r} per step, this run by {flat.gamma!r}")
    n = int(scheduler_state_dict.get("last_epoch", 0))
    steps = {int(s["step"]) for s in opt.state.values() if "step" in s}
    if steps and steps != {n}:
        raise ValueError(f"optimizer steps {sorted(steps)} do not match scheduler.last_epoch = {n}")
    with torch.no_grad():
        full = (torch.zeros(flat.padded, device=flat.flat_p.device), torch.zeros(flat.padded, device=flat.flat_p.device))
        for p, (m, v) in zip(flat.params, flat.moment_views(full)):
            st = opt.state.get(p)
            if st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
        flat.load_moments(*full)      # with sharded moments (ZeRO-1) every rank keeps its own slice
    flat.set_state(n, [float(g.get("initial_lr", g["lr"] / gamma ** n)) for g in opt.param_groups])
    return n


def _gather_rng_states(trainer):
    """[world, 3] int64 on the CPU: every rank's (seed, counter, scratch) of the model's device-resident Philox stream, or None when the model has
    not drawn yet.  Ranks reseed after the common initialisation (trainer.py: seed + 7919 (rank + 1)) so that their ray jitter, inverse-CDF and
    Eikonal draws differ; a checkpoint that kept only the writing rank's stream would hand that ONE stream to every rank of the resumed run."""
    rng = getattr(trainer.model, "_rng_state", None)
    world = int(getattr(trainer, "world_size", 1) or 1)
    if world > 1 and getattr(trainer, "dp", False):
        import torch.distributed as dist
        mine = (torch.full((3,), -1, dtype=torch.int64) if rng is None else rng.detach().cpu().to(torch.int64)).to(trainer.device if dist.get_backend() == "nccl" else "cpu")
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        allr = torch.stack([r.cpu() for r in rows])
        return None if bool((allr[:, 0] < 0).any()) else allr
    return None if rng is None else rng.detach().cpu().to(torch.int64).view(1, 3)


def _restore_rng_state(trainer, saved):
    """Row `rank` of a [world, 3] state saved at the same world size; anything else (another world size, the one-row form of an older checkpoint
    loaded by several ranks) leaves the streams the ranks seeded for themselves: correlated draws across ranks would be worse than a restart."""
    saved = saved.view(-1, 3)
    world, rank = int(getattr(trainer, "world_size", 1) or 1), int(getattr(trainer, "rank", 0) or 0)
    if saved.shape[0] != world:
        import warnings
        warnings.warn(f"checkpoint holds {saved.shape[0]} draw stream(s), this run has {world} rank(s): keeping the freshly seeded per-rank streams")
        return
    trainer.model.rng_state(trainer.device).copy_(saved[rank].to(trainer.device))


def save_checkpoints(trainer, checkpoints_path, epoch, write=True):
    """Under data parallelism EVERY rank calls this (the ZeRO-1 moment gather is a collective); pass write=(rank == 0)."""
    opt_sd, sched_sd = optimizer_state_dicts(trainer)
    rng = _gather_rng_states(trainer)                      # a collective under data parallelism: before the non-writing ranks leave
    if not write:
        return
    model_payload = {"epoch": epoch, "model_state_dict": trainer.model.state_dict()}
    if rng is not None:         # the device-resident Philox streams of the iteration's draws (hs_iter_prologue), one row per rank: an extra key
        model_payload["hs_rng_state"] = rng                # the reference's loader ignores; without it a resumed run restarts the draw stream
    payloads = {MODEL_DIR: model_payload,
                OPTIMIZER_DIR: {"epoch": epoch, "optimizer_state_dict": opt_sd},
                SCHEDULER_DIR: {"epoch": epoch, "scheduler_state_dict": sched_sd}}
    for sub, payload in payloads.items():
        os.makedirs(os.path.join(checkpoints_path, sub), exist_ok=True)
        for name in (str(epoch), "latest"):
            torch.save(payload, os.path.join(checkpoints_path, sub, name + ".pth"))


def load_checkpoints(trainer, checkpoints_path, checkpoint="latest", map_location=None):
    """Returns the stored epoch.  Model keys may carry DataParallel's 'module.' prefix (holoscene_train.py:182-185)."""
    dev = map_location or trainer.device
    saved = torch.load(os.path.join(checkpoints_path, MODEL_DIR, str(checkpoint) + ".pth"), map_location=dev)
    trainer.model.load_state_dict({k.replace("module.", ""): v for k, v in saved["model_state_dict"].items()})
    if "hs_rng_state" in saved and hasattr(trainer.model, "rng_state"):
        _restore_rng_state(trainer, saved["hs_rng_state"])
    opt = torch.load(os.path.join(checkpoints_path, OPTIMIZER_DIR, str(checkpoint) + ".pth"), map_location=dev)
    sched = torch.load(os.path.join(checkpoints_path, SCHEDULER_DIR, str(checkpoint) + ".pth"), map_location=dev)
    trainer.iter_step = load_optimizer_state(trainer, opt["optimizer_state_dict"], sched["scheduler_state_dict"])
    if hasattr(trainer.model.implicit_network, "invalidate_packed_weights"):
        trainer.model.implicit_network.invalidate_packed_weights()
    return saved["epoch"]
