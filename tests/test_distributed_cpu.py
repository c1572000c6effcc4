"""CPU, world_size 2, gloo: the frame-sharded data-parallel exchange (holoscene_amd/training/distributed.py).

Parity definition: after the exchange every rank holds the mean of the ranks' gradients, so one
Adam step equals a single process that averaged the per-rank gradients itself."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from holoscene_amd.training.distributed import average_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must align them
    big = torch.nn.Parameter(torch.randn(1 << 20, 2))       # above the coalescing threshold, reduced in place
    small = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(()))]
    holder = torch.nn.ParameterList([big] + small)
    broadcast_parameters(holder, src=0)
    start = [p.detach().clone() for p in holder]
    g = torch.Generator().manual_seed(7 + rank)
    for p in holder:
        p.grad = torch.randn(p.shape, generator=g)
    local = [p.grad.clone() for p in holder]
    average_gradients(holder, world)
    opt = torch.optim.Adam(holder, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    opt.step()
    q.put((rank, [t.numpy() for t in start], [t.numpy() for t in local], [p.grad.numpy() for p in holder], [p.detach().numpy() for p in holder]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_mean_and_adam_step_match_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, start0, loc0, avg0, new0), (_, start1, loc1, avg1, new1) = res
    for a, b in zip(start0, start1):
        assert (a == b).all(), "broadcast_parameters must align the replicas"
    for l0, l1, a0, a1 in zip(loc0, loc1, avg0, avg1):
        mean = (torch.from_numpy(l0) + torch.from_numpy(l1)) / 2
        assert torch.allclose(torch.from_numpy(a0), mean, rtol=1e-6, atol=1e-7)
        assert (a0 == a1).all(), "every rank must hold the same averaged gradient"
    # single-process reference: average the two local gradients, one Adam step
    ref = [torch.nn.Parameter(torch.from_numpy(s.copy())) for s in start0]
    for p, l0, l1 in zip(ref, loc0, loc1):
        p.grad = (torch.from_numpy(l0) + torch.from_numpy(l1)) / 2
    torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-15).step()
    for p, n0, n1 in zip(ref, new0, new1):
        assert torch.allclose(p.detach(), torch.from_numpy(n0), rtol=1e-6, atol=1e-7)
        assert (n0 == n1).all()
