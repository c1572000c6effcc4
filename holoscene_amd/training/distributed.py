"""Frame-sharded data parallelism.  The reference is single-GPU; this is new.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
Every rank renders its own frame/rays; the only exchange is the parameter-gradient mean:
  * the two 48.8 MB hash-grid gradients are reduced in place, one collective each (large payloads
    ride RCCL's direct/ring algorithms at link rate; no bucketing copy);
  * the ~40 small MLP/beta gradients are flattened into one buffer and reduced with one collective.
Parity definition: the result equals one process averaging the gradients of the ranks' independent
single-frame iterations, then taking one Adam step.
"""
import torch
import torch.distributed as dist

_SMALL = 1 << 20  # elements; tensors below this are coalesced


def average_gradients(params, world_size, group=None):
    grads = [p.grad for p in params if p.grad is not None]
    big = [g for g in grads if g.numel() >= _SMALL]
    small = [g for g in grads if g.numel() < _SMALL]
    handles = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in big]
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world_size)
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for h in handles:
        h.wait()
    for g in big:
        g.mul_(1.0 / world_size)


def broadcast_parameters(module, src=0, group=None):
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def reduce_scatter_sum(out_shard, full, world_size, rank, group=None):
    """out_shard <- sum over ranks of full[rank*n:(rank+1)*n].  RCCL: one reduce_scatter_tensor, out of place (a staging slice of
    1/N of the buffer: no reliance on the backend's in-place aliasing rules).  gloo (CPU tests) has no reduce-scatter: the same
    sums through one `reduce` per destination rank over that rank's slice -- identical shard arithmetic, so the world-size-2
    CPU tests cover exactly the indexing the RCCL branch uses."""
    n = out_shard.numel()
    if dist.get_backend(group) == "gloo":
        for r in range(world_size):     # `dst` is a GLOBAL rank: translate the group-local index when a sub-group is used
            dst = r if group is None else dist.get_global_rank(group, r)
            dist.reduce(full[r * n:(r + 1) * n], dst=dst, op=dist.ReduceOp.SUM, group=group)    # only dst's slice is defined afterwards
        out_shard.copy_(full[rank * n:(rank + 1) * n])
    else:
        dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=group)


def exchange_segment(flat, s, world_size, group=None):
    """ZeRO-1 on segment s of the flat buffers (training/flat.py): reduce-scatter its gradient (each rank receives the SUM of its
    1/N slice in a staging slice), fused Adam on that slice straight from the staging slice (gradient scaled by 1/N inside the
    kernel), all-gather of the updated parameter slices.  `flat.tick()` must have run for this update.  Everything is enqueued on
    the CURRENT stream: the trainer calls this for the early segment on a side stream while the backward pass is still running."""
    b, e = flat.shards[s]
    sb, se = flat.segments[s]
    shard_g = flat.shard_grad(s)
    reduce_scatter_sum(shard_g, flat.flat_g[sb:se], world_size, flat.rank, group)
    flat.step_segment(s, grad_scale=1.0 / world_size, grad_shard=shard_g)
    send = flat.shard_send(s)
    send.copy_(flat.flat_p[b:e])
    dist.all_gather_into_tensor(flat.flat_p[sb:se], send, group=group)


def exchange_and_step_flat(flat, world_size, zero1=True, group=None, done=()):
    """Data-parallel step over flat buffers (training/flat.py).

    zero1=True (default): per segment of the buffer, reduce-scatter -> shard-local fused Adam -> all-gather (exchange_segment).
    Per rank this moves (N-1)/N of the buffer twice -- the same volume as a ring all-reduce -- and cuts the optimiser's HBM
    traffic by N; with FlatAdam(shard_moments=True) (what Stage1Trainer builds for zero1) also the moment storage.
    Element-for-element identical to "all-reduce mean, full Adam".  done: segments the caller has already exchanged in this update
    (the trainer's overlapped early segment).
    zero1=False: one all-reduce over the flat gradient, full Adam on every rank."""
    if zero1:
        flat.tick()
        for s in range(len(flat.segments)):
            if s not in done:
                exchange_segment(flat, s, world_size, group)
        flat.end_update()
    else:
        dist.all_reduce(flat.flat_g, op=dist.ReduceOp.SUM, group=group)
        flat.step(grad_scale=1.0 / world_size)
