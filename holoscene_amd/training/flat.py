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
    def __init__(self, model, lr, lr_factor_for_grid, decay_rate, decay_steps, betas=(0.9, 0.99), eps=1e-15, world_size=1, rank=0,
                 shard_moments=False):
        """shard_moments (ZeRO-1): this rank stores the Adam moments of its own 1/world_size slice only; `step` must then be
        called with shard_only=True and checkpoints gather the slices (`gather_moments`)."""
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
        shard = self.padded // world_size
        self.shard = (rank * shard, (rank + 1) * shard)
        self.shard_moments = bool(shard_moments) and world_size > 1
        self.mv_base = self.shard[0] if self.shard_moments else 0       # flat index of flat_m[0] / flat_v[0]
        self.flat_m = torch.zeros(shard if self.shard_moments else self.padded, device=dev)
        self.flat_v = torch.zeros(shard if self.shard_moments else self.padded, device=dev)
        self._shard_g = None            # staging slice the reduce-scatter writes (distributed.py)
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
        """Dense-update semantics: the fused Adam walks the WHOLE flat buffer, so a parameter that received no gradient this
        iteration is updated with a zero gradient (its moments decay, its momentum still moves it), whereas torch.optim.Adam after
        `zero_grad(set_to_none=True)` -- the reference's loop -- skips it.  Stage 1 uses every parameter in every iteration, so the
        two agree there; a model with conditionally used parameters is told once."""
        src = [p.grad for p, _ in self.small if p.grad is not None]
        dst = [v for p, v in self.small if p.grad is not None]
        if len(src) != len(self.small) and not getattr(self, "_warned_missing_grad", False):
            import warnings
            self._warned_missing_grad = True
            warnings.warn(f"FlatAdam: {len(self.small) - len(src)} of {len(self.small)} small parameters received no gradient; the flat "
                          "optimiser updates them with a zero gradient (torch.optim.Adam would skip them)")
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in self.small:
            p.grad = v

    def moment_views(self, full=None):
        """[(exp_avg, exp_avg_sq)] per parameter, views into the flat moment buffers, in optimiser order.
        full: (m, v) full-length buffers -- required when this rank holds only its shard (see gather_moments)."""
        if full is None:
            if self.shard_moments:
                raise RuntimeError("moments are sharded across ranks (ZeRO-1): pass full=gather_moments()")
            full = (self.flat_m, self.flat_v)
        fm, fv = full
        out, off = [], 0
        for p in self.params:
            n = p.numel()
            out.append((fm[off:off + n].view_as(p), fv[off:off + n].view_as(p)))
            off += n
        return out

    def gather_moments(self, group=None):
        """Full-length (m, v).  ZeRO-1 leaves every rank with valid moments for its own slice only -- whether the buffers are
        shard-sized (shard_moments) or full-sized and stepped with shard_only=True -- so an export must collect the slices:
        a COLLECTIVE call (all ranks) when world_size > 1."""
        if self.world_size == 1:
            return self.flat_m, self.flat_v
        import torch.distributed as dist
        b, e = self.shard
        out = []
        for t in (self.flat_m, self.flat_v):
            mine = (t if self.shard_moments else t[b:e]).contiguous().clone()
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            dist.all_gather(parts, mine, group=group)
            out.append(torch.cat(parts))
        return tuple(out)

    def load_moments(self, full_m, full_v):
        """Inverse of gather_moments: every rank keeps (at least) its own slice."""
        b, e = self.shard
        if self.shard_moments:
            self.flat_m.copy_(full_m[b:e])
            self.flat_v.copy_(full_v[b:e])
        else:
            self.flat_m.copy_(full_m)
            self.flat_v.copy_(full_v)

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
    def shard_grad(self):
        """Staging slice for the reduce-scatter output (1/world_size of the flat gradient)."""
        if self._shard_g is None:
            b, e = self.shard
            self._shard_g = torch.empty(e - b, device=self.flat_g.device, dtype=self.flat_g.dtype)
        return self._shard_g

    def step(self, grad_scale=1.0, shard_only=False, grad_shard=None):
        """One Adam + ExponentialLR step.  shard_only=True updates this rank's slice only (ZeRO-1); grad_shard: that slice's
        summed gradient in its own buffer (the reduce-scatter output) instead of flat_g[shard]."""
        be = _be._backend
        if self.shard_moments and not shard_only:
            raise RuntimeError("this rank stores only its shard of the Adam moments: step(shard_only=True)")
        be.adam_tick(self.state, self.betas[0], self.betas[1], self.gamma)
        b, e = self.shard if shard_only else (0, self.padded)
        g, g_base = (self.flat_g, 0) if grad_shard is None else (grad_shard, self.shard[0])
        be.adam_flat(self.flat_p, g, self.flat_m, self.flat_v, b, e, self.state, self.betas[0], self.betas[1], self.eps, grad_scale,
                     g_base=g_base, mv_base=self.mv_base)

    def state_dict(self):
        """This rank's optimiser state (with sharded moments: its slice)."""
        return {"flat_m": self.flat_m, "flat_v": self.flat_v, "state": self.state}

    def load_state_dict(self, sd):
        self.flat_m.copy_(sd["flat_m"])
        self.flat_v.copy_(sd["flat_v"])
        self.state.copy_(sd["state"])
