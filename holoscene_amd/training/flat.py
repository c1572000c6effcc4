"""Flat parameter / gradient storage and the fused Adam step.

All trainable tensors of the Stage-1 model become views into ONE fp32 buffer, ordered by the reference's optimiser
groups (training/holoscene_train.py:156-164): hash grids | MLPs | beta.  Gradients are views into a second flat
buffer that stays attached (``p.grad``), so
  * ``zero_grad`` is one memset,
  * the data-parallel exchange is one collective over one contiguous buffer (or reduce-scatter / all-gather
    around a shard-local Adam: ZeRO-1),
  * Adam is one streaming kernel (csrc/optim.hip) -- no host work, capturable in a HIP graph.
State-dict names and shapes are untouched (the views keep their module attributes).
"""
import ctypes

import torch

from ..hashencoder import backend as _be


class FlatAdam:
    def __init__(self, model, lr, lr_factor_for_grid, decay_rate, decay_steps, betas=(0.9, 0.99), eps=1e-15, world_size=1, rank=0):
        groups = [list(model.implicit_network.grid_parameters()),
                  list(model.implicit_network.mlp_parameters()) + list(model.rendering_network.parameters()),
                  list(model.density.parameters())]
        self.params = [p for g in groups for p in g]
        dev = self.params[0].device
        sizes = [sum(p.numel() for p in g) for g in groups]
        total = sum(sizes)
        align = 4 * world_size                      # every rank's shard is a whole number of 16-byte quads
        self.numel = total
        self.padded = (total + align - 1) // align * align
        self.flat_p = torch.zeros(self.padded, device=dev)
        self.flat_g = torch.zeros(self.padded, device=dev)
        self.flat_m = torch.zeros(self.padded, device=dev)
        self.flat_v = torch.zeros(self.padded, device=dev)
        off = 0
        self.small = []          # (parameter, gradient view) of everything that is not a hash table
        n_tables = len(groups[0])
        for i, p in enumerate(self.params):
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            p.grad = self.flat_g[off:off + n].view_as(p)
            if i >= n_tables:
                self.small.append((p, p.grad))
            else:
                p._hs_flat_owner = True   # its gradient is consumed through gather_grads() only: scatters may use the side stream
            off += n
        self.betas, self.eps = betas, eps
        self.gamma = float(decay_rate) ** (1.0 / float(decay_steps))
        self.world_size, self.rank = world_size, rank
        shard = self.padded // world_size
        self.shard = (rank * shard, (rank + 1) * shard)
        st = _be.hsAdamState()
        st.step = 0
        st.group_end[0], st.group_end[1] = sizes[0], sizes[0] + sizes[1]
        for i, v in enumerate((lr * lr_factor_for_grid, lr, lr)):
            st.lr0[i] = v
            st.lr[i] = v
        self.state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(dev)

    # ---- gradients
    def zero_grad(self):
        """One memset; the hash tables keep their gradient views attached (the scatter kernels accumulate into them in
        place), the ~30 small MLP tensors are detached so that autograd hands over each gradient tensor as is instead of
        launching one `grad += new` kernel per parameter -- `gather_grads` then moves them with one multi-tensor copy."""
        self.flat_g.zero_()
        for p, _ in self.small:
            p.grad = None

    def gather_grads(self):
        src = [p.grad for p, _ in self.small if p.grad is not None]
        dst = [v for p, v in self.small if p.grad is not None]
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in self.small:
            p.grad = v

    def moment_views(self):
        """[(exp_avg, exp_avg_sq)] per parameter, views into the flat moment buffers, in optimiser order."""
        out, off = [], 0
        for p in self.params:
            n = p.numel()
            out.append((self.flat_m[off:off + n].view_as(p), self.flat_v[off:off + n].view_as(p)))
            off += n
        return out

    def set_state(self, step, lr0):
        """Resume: `step` updates done so far, lr0 = the three groups' initial learning rates (checkpoint.py)."""
        st = self.read_state()
        st.step = int(step)
        for i, v in enumerate(lr0):
            st.lr0[i] = v
            st.lr[i] = v * self.gamma ** max(int(step) - 1, 0)
        self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))

    def read_state(self):
        return _be.hsAdamState.from_buffer_copy(bytes(self.state.cpu().numpy().tobytes()))

    # ---- update
    def step(self, grad_scale=1.0, shard_only=False):
        """One Adam + ExponentialLR step.  shard_only=True updates this rank's slice only (ZeRO-1)."""
        be = _be._backend
        be.adam_tick(self.state, self.betas[0], self.betas[1], self.gamma)
        b, e = self.shard if shard_only else (0, self.padded)
        be.adam_flat(self.flat_p, self.flat_g, self.flat_m, self.flat_v, b, e, self.state, self.betas[0], self.betas[1], self.eps, grad_scale)

    def state_dict(self):
        return {"flat_m": self.flat_m, "flat_v": self.flat_v, "state": self.state}

    def load_state_dict(self, sd):
        self.flat_m.copy_(sd["flat_m"])
        self.flat_v.copy_(sd["flat_v"])
        self.state.copy_(sd["state"])
