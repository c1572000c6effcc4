"""Flat parameter / gradient storage and the fused Adam step.

All trainable tensors of the Stage-1 model become views into ONE fp32 buffer, ordered by the reference's optimiser
groups (training/holoscene_train.py:156-164): hash grids | MLPs | beta.  Gradients are views into a second flat
buffer that stays attached (``p.grad``), so
  * ``zero_grad`` is one memset,
  * the data-parallel exchange is one collective over one contiguous buffer (or reduce-scatter / all-gather
    around a shard-local Adam: ZeRO-1),
  * Adam is one streaming kernel (csrc/optim.hip) -- no host work, capturable in a HIP graph.
State-dict names and shapes are untouched (the views keep their module attributes).

Segments (data parallelism only).  The exchange of a parameter can start as soon as ITS gradient is final, and the colour table's is
final 0.6-0.75 ms before the iteration's last kernel (its scatter runs right after the appearance backward; the trunk backward,
the SDF table's scatter and the trunk's weight-gradient GEMMs follow), the SDF table's ~0.2 ms before it.  ``early_params``
therefore splits the buffer into independently sharded segments, [early table 0 | pad] [early table 1 | pad] ... [everything
else | pad]: each is reduce-scattered / stepped / all-gathered on its own (training/distributed.py::exchange_segment), the early
ones on a side stream under the rest of the backward pass (trainer.py).  Each segment is a whole number of 16-byte quads per
rank; the pad elements are zero parameters with zero gradients.  Without ``early_params`` there is one segment and the layout is
the plain concatenation.
"""
import contextlib
import ctypes

import torch

from ..hashencoder import backend as _be


class FlatAdam:
    ZERO_POOL = 8192        # floats of zeroed scratch behind the gradients (backend.zeros_small: bias-gradient accumulators of the backward kernels)

    def __init__(self, model, lr, lr_factor_for_grid, decay_rate, decay_steps, betas=(0.9, 0.99), eps=1e-15, world_size=1, rank=0,
                 shard_moments=False, early_params=None):
        """shard_moments (ZeRO-1): this rank stores the Adam moments of its own 1/world_size slice (of every segment) only; the
        update then goes through `tick` + `step_segment` and checkpoints gather the slices (`gather_moments`).
        early_params: hash tables (members of the first optimiser group) in the order in which their gradients become final in the
        backward pass; they are laid out first, each as a segment of its own (module docstring)."""
        groups = [list(model.implicit_network.grid_parameters()),
                  list(model.implicit_network.mlp_parameters()) + list(model.rendering_network.parameters()),
                  list(model.density.parameters())]
        early = list(early_params or [])
        early_ids = {id(p) for p in early}
        if early_ids - {id(p) for p in groups[0]}:
            raise ValueError("early_params must be hash tables of the first optimiser group")
        groups[0] = early + [p for p in groups[0] if id(p) not in early_ids]
        self.params = [p for g in groups for p in g]
        dev = self.params[0].device
        align = 4 * world_size                      # every rank's shard of every segment is a whole number of 16-byte quads
        up = lambda n: (n + align - 1) // align * align  # noqa: E731
        self.offsets, off, cuts = [], 0, [0]        # flat index of each parameter's first element; segment boundaries
        n_tables = len(groups[0])
        for i, p in enumerate(self.params):
            if 0 < i <= len(early):                 # a boundary after every early table
                off = up(off)
                cuts.append(off)
            if i <= n_tables:                       # every table, and what follows the last one, starts a 16-byte quad: the kernels that
                off = (off + 3) // 4 * 4            # step a table on its own (k_hash_bin_step, hs_adam_flat over a sub-range) walk whole quads
            self.offsets.append(off)
            off += p.numel()
        self.numel = off                            # end of the last parameter (flat index space, inner pads included)
        self.padded = up(off)
        if cuts[-1] != self.padded:
            cuts.append(self.padded)
        self.segments = list(zip(cuts[:-1], cuts[1:]))
        self.flat_p = torch.zeros(self.padded, device=dev)
        self._g_alloc = torch.zeros(self.padded + self.ZERO_POOL, device=dev)     # gradients | pool of small zeroed accumulators (backend.zeros_small)
        self.flat_g = self._g_alloc[:self.padded]
        self.shards = [(b + rank * ((e - b) // world_size), b + (rank + 1) * ((e - b) // world_size)) for b, e in self.segments]
        self.shard_moments = bool(shard_moments) and world_size > 1
        # moment storage: full length, or this rank's slices of the segments back to back; mv_bases[s] = flat index that
        # element 0 of the moment buffers would have for segment s (the kernel indexes m[i - mv_base])
        held = sum(e - b for b, e in self.shards)
        self.flat_m = torch.zeros(held if self.shard_moments else self.padded, device=dev)
        self.flat_v = torch.zeros(held if self.shard_moments else self.padded, device=dev)
        self.mv_bases, o = [], 0
        for b, e in self.shards:
            self.mv_bases.append(b - o if self.shard_moments else 0)
            o += e - b
        self._shard_g = {}              # per segment: staging slice the reduce-scatter writes (distributed.py)
        self._shard_p = {}              # per segment: send buffer of the all-gather
        self._ticked = False            # tick() already ran for the update in flight (segment-wise stepping)
        self.small = []          # (parameter, gradient view) of everything that is not a hash table
        sizes_end = [0, 0]
        for i, p in enumerate(self.params):
            n, off = p.numel(), self.offsets[i]
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            p.grad = self.flat_g[off:off + n].view_as(p)
            if i >= n_tables:
                self.small.append((p, p.grad))
                p._hs_flat_view = p.grad    # producers that can write a gradient in place do (model/network.py: flat_grad_view)
            else:
                p._hs_flat_owner = True   # its gradient is consumed through gather_grads() only: the scatters write it in place
            if i < n_tables:
                sizes_end[0] = off + n
            if i < n_tables + len(groups[1]):
                sizes_end[1] = off + n
        self.n_tables, self.tables_end = n_tables, sizes_end[0]
        self._stepped = []              # flat ranges whose Adam step rode on their scatter in the pass that just ended (table_steps)
        # direct gradient writes (model/network.py: flat_grad_target): which small parameters' views a backward kernel has written in the pass
        # that is open -- between zero_grad() and gather_grads() --, by parameter identity.  Owned here, not process-wide: a second optimiser,
        # model.zero_grad() or a backward pass that was not preceded by zero_grad() cannot meet a stale claim (the pass is then not open and
        # every producer hands its gradient to autograd the ordinary way)
        self.claims, self.pass_open = {}, False
        for p, _ in self.small:
            p._hs_flat = self
        self.betas, self.eps = betas, eps
        self.gamma = float(decay_rate) ** (1.0 / float(decay_steps))
        self.world_size, self.rank = world_size, rank
        st = _be.hsAdamState()
        st.step = 0
        st.group_end[0], st.group_end[1] = sizes_end[0], max(sizes_end)
        for i, v in enumerate((lr * lr_factor_for_grid, lr, lr)):
            st.lr0[i] = v
            st.lr[i] = v
        self.state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(dev)

    # ---- gradients
    def zero_grad(self, tables=True, defer=False):
        """One memset; the hash tables keep their gradient views attached (the scatter kernels accumulate into them in
        place), the ~30 small MLP tensors are detached so that autograd hands over each gradient tensor as is instead of
        launching one `grad += new` kernel per parameter -- `gather_grads` then moves them with one multi-tensor copy.
        tables=False: the tables' gradient storage is known to be all zero (it was never written since the allocation, or the last
        pass ended with `table_steps` / `clear_table_grads`): only the small tensors and the pool behind them are cleared.
        defer (with tables=False): do not launch that memset but RETURN the range; the caller has it cleared by the iteration's first kernel."""
        deferred = None
        if tables:
            self._g_alloc.zero_()
        elif defer:     # the caller clears this range itself before any gradient is produced (hs_iter_prologue's zero range: no launch of its own)
            first_small = self.offsets[self.n_tables] if self.n_tables < len(self.offsets) else self.tables_end     # (no small parameter at all: the pool only)
            deferred = self._g_alloc[first_small:]        # (the first small parameter starts a 16-byte quad; the pads before it are never written)
        else:
            self._g_alloc[self.tables_end:].zero_()
        _be.set_zero_pool(self._g_alloc[self.padded:])
        self.claims.clear()              # no view has been written yet: the first producer of each writes, later ones accumulate
        self.pass_open = True
        for p, _ in self.small:
            p.grad = None
        return deferred

    def gather_grads(self):
        """Dense-update semantics: the fused Adam walks the WHOLE flat buffer, so a parameter that received no gradient this
        iteration is updated with a zero gradient (its moments decay, its momentum still moves it), whereas torch.optim.Adam after
        `zero_grad(set_to_none=True)` -- the reference's loop -- skips it.  Stage 1 uses every parameter in every iteration, so the
        two agree there; a model with conditionally used parameters is told once."""
        _be.set_zero_pool(None)         # the backward pass this pool served is over
        self.pass_open = False
        for p in self.claims.values():  # a view a kernel wrote directly was handed to autograd as that parameter's gradient: it ends the pass as
            if p.grad is None:          # .grad, or inside a sum autograd formed with another producer's tensor (copied back below) -- never as nothing
                raise RuntimeError("FlatAdam: a backward kernel wrote a parameter's gradient straight into the flat buffer, but autograd did not adopt "
                                   "it: the parameter ends the backward pass without a .grad (a gradient was dropped)")
        self.claims.clear()
        have = [(p.grad, v) for p, v in self.small if p.grad is not None]
        if len(have) != len(self.small) and not getattr(self, "_warned_missing_grad", False):
            import warnings
            self._warned_missing_grad = True
            warnings.warn(f"FlatAdam: {len(self.small) - len(have)} of {len(self.small)} small parameters received no gradient; the flat "
                          "optimiser updates them with a zero gradient (torch.optim.Adam would skip them)")
        # a producer that wrote its result straight into the view (hs_iter_epilogue, hs_assemble with a destination) needs no copy
        src = [g for g, v in have if g.data_ptr() != v.data_ptr()]
        dst = [v for g, v in have if g.data_ptr() != v.data_ptr()]
        if src:
            if src[0].is_cuda and all(t.is_contiguous() and t.dtype == torch.float32 for t in src):
                _be._backend.copy_many(dst, src)        # one launch, one element per thread (the multi-tensor copy is 17 us here)
            else:
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
        return [(fm[off:off + p.numel()].view_as(p), fv[off:off + p.numel()].view_as(p)) for p, off in zip(self.params, self.offsets)]

    def _held(self, t, s):
        """This rank's slice of segment s inside a moment buffer (shard-sized or full-length storage)."""
        b, e = self.shards[s]
        return t[b - self.mv_bases[s]:e - self.mv_bases[s]]

    def gather_moments(self, group=None):
        """Full-length (m, v).  ZeRO-1 leaves every rank with valid moments for its own slices only -- whether the buffers are
        shard-sized (shard_moments) or full-sized and stepped segment-wise -- so an export must collect the slices:
        a COLLECTIVE call (all ranks) when world_size > 1."""
        if self.world_size == 1:
            return self.flat_m, self.flat_v
        import torch.distributed as dist
        out = []
        for t in (self.flat_m, self.flat_v):
            full = torch.zeros(self.padded, device=t.device, dtype=t.dtype)
            for s, (b, e) in enumerate(self.segments):
                mine = self._held(t, s).contiguous().clone()
                parts = [torch.empty_like(mine) for _ in range(self.world_size)]
                dist.all_gather(parts, mine, group=group)
                full[b:e] = torch.cat(parts)
            out.append(full)
        return tuple(out)

    def load_moments(self, full_m, full_v):
        """Inverse of gather_moments: every rank keeps (at least) its own slices."""
        if self.shard_moments:
            for s, (b, e) in enumerate(self.shards):
                self._held(self.flat_m, s).copy_(full_m[b:e])
                self._held(self.flat_v, s).copy_(full_v[b:e])
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
    def shard_grad(self, s=0):
        """Staging slice for the reduce-scatter output of segment s (1/world_size of the segment's gradient)."""
        if s not in self._shard_g:
            b, e = self.shards[s]
            self._shard_g[s] = torch.empty(e - b, device=self.flat_g.device, dtype=self.flat_g.dtype)
        return self._shard_g[s]

    def shard_send(self, s=0):
        """Send buffer of segment s's all-gather (the updated parameter slice, copied out of the buffer the gather writes)."""
        if s not in self._shard_p:
            b, e = self.shards[s]
            self._shard_p[s] = torch.empty(e - b, device=self.flat_p.device, dtype=self.flat_p.dtype)
        return self._shard_p[s]

    def tick(self):
        """Advance the step count, bias corrections and learning rates (ExponentialLR) ONCE per update; the segment steps of that
        update then all read the same state.  Idempotent until `end_update`."""
        if not self._ticked:
            _be._backend.adam_tick(self.state, self.betas[0], self.betas[1], self.gamma)
            self._ticked = True

    def end_update(self):
        self._ticked = False

    def step_segment(self, s, grad_scale=1.0, grad_shard=None):
        """Adam on this rank's slice of segment s (ZeRO-1); grad_shard: that slice's summed gradient in its own buffer (the
        reduce-scatter output) instead of flat_g[slice].  `tick()` must have run for this update."""
        if not self._ticked:
            raise RuntimeError("step_segment before tick()")
        b, e = self.shards[s]
        g, g_base = (self.flat_g, 0) if grad_shard is None else (grad_shard, b)
        _be._backend.adam_flat(self.flat_p, g, self.flat_m, self.flat_v, b, e, self.state, self.betas[0], self.betas[1], self.eps, grad_scale,
                               g_base=g_base, mv_base=self.mv_bases[s])

    def step(self, grad_scale=1.0):
        """One Adam + ExponentialLR step over the whole buffer (single process, or after an all-reduce of the gradient) -- minus the
        tables that took their step inside their scatter during the backward pass that just ended (`table_steps`)."""
        if self.shard_moments:
            raise RuntimeError("this rank stores only its shards of the Adam moments: tick() + step_segment(s)")
        self.tick()
        at = 0
        for b, e in sorted(self._stepped) + [(self.padded, self.padded)]:
            if b > at:
                _be._backend.adam_flat(self.flat_p, self.flat_g, self.flat_m, self.flat_v, at, b, self.state, self.betas[0], self.betas[1],
                                       self.eps, grad_scale, g_base=0, mv_base=0)
            at = max(at, e)
        self._stepped = []
        self.end_update()

    # ---- reduce-and-step: the tables' Adam update inside their scatter's reduction (csrc/hash_encode.hip: k_hash_bin_step)
    def table_steps_supported(self):
        """Single process with full-length moments (every table starts a 16-byte quad and is followed by zero padding up to the next one:
        the rest of the buffer is stepped by hs_adam_flat over quad-aligned ranges)."""
        return self.world_size == 1 and not self.shard_moments and self.flat_p.is_cuda and self.n_tables > 0

    @contextlib.contextmanager
    def table_steps(self, grad_scale=1.0, producers=None):
        """For the backward pass inside the block, every hash table whose gradient has exactly ONE producer (a scatter through
        backend.bwd / bwd_jac into its flat gradient view) takes its Adam step in that producer's reduction kernel.  Needs: `tick()`
        done for this update (the iteration prologue), the tables' gradient storage all zero on entry (it is again on exit).  A table
        that received no scatter keeps its turn in `step()`; the following `step()` covers exactly what is left.
        producers: per table (optimiser order) the number of scatters this pass will send it (default 1 each); with n > 1 the first n - 1
        accumulate into the gradient table the plain way and the last one steps, adding what it finds there (hsTableStep.prior)."""
        if not self._ticked:
            raise RuntimeError("table_steps before tick(): the reduction kernels read the advanced optimiser state")
        mine = {}
        for i in range(self.n_tables):
            p, off = self.params[i], self.offsets[i]
            n_prod = 1 if producers is None else int(producers[i])
            ts = _be.hsTableStep(p.data.data_ptr(), self.flat_m[off:].data_ptr(), self.flat_v[off:].data_ptr(), self.state.data_ptr(),
                                 self.betas[0], self.betas[1], self.eps, grad_scale, 0, 1 if n_prod > 1 else 0)
            key = self.flat_g[off:].data_ptr()
            mine[key] = (off, (off + p.numel() + 3) // 4 * 4)       # (the pad behind a table: zero parameters with zero gradients)
            _be.TABLE_STEPS[key] = [ts, 0, n_prod]
        try:
            yield
        finally:
            done = {k: _be.TABLE_STEPS.pop(k) for k in mine}
            short = [k for k, e in done.items() if 0 < e[1] < e[2]]
            self._stepped = [mine[k] for k, e in done.items() if e[1] >= e[2]]
            if short:       # fewer producers than counted: their contributions sit in the gradient table and no one has stepped the table
                raise RuntimeError("reduce-and-step: a hash table received fewer gradient producers than were counted for this variant")

    def clear_table_grads(self):
        """Leave the tables' gradient storage all zero (the end of a pass that accumulated into it the plain way, in a trainer whose
        other passes rely on `zero_grad(tables=False)`)."""
        self.flat_g[:self.tables_end].zero_()

    LAYOUT_VERSION = 2      # 2: every table (and what follows the last one) starts a 16-byte quad (round 4); 1: plain concatenation

    def layout(self):
        return {"version": self.LAYOUT_VERSION, "offsets": list(self.offsets), "numel": int(self.numel), "padded": int(self.padded),
                "segments": [list(s) for s in self.segments], "shards": [list(s) for s in self.shards], "shard_moments": bool(self.shard_moments)}

    def state_dict(self):
        """This rank's optimiser state (with sharded moments: its slice) and the layout it is laid out in: the raw buffers mean nothing
        without the offsets (for an exchange format use training/checkpoint.py, which goes through per-parameter tensors)."""
        return {"flat_m": self.flat_m, "flat_v": self.flat_v, "state": self.state, "layout": self.layout()}

    def load_state_dict(self, sd):
        lay = sd.get("layout")
        if lay is None:
            if tuple(sd["flat_m"].shape) != tuple(self.flat_m.shape):
                raise ValueError("FlatAdam.load_state_dict: the state carries no layout record and its length differs from this optimiser's; "
                                 "it was written by another layout version -- reload through training/checkpoint.py (per-parameter format)")
            import warnings
            warnings.warn("FlatAdam.load_state_dict: state without a layout record (written before layout version 2): lengths match, offsets unchecked")
        else:
            mine = self.layout()
            for k in ("version", "offsets", "padded", "shards", "shard_moments"):
                if lay.get(k) != mine[k]:
                    raise ValueError(f"FlatAdam.load_state_dict: layout mismatch in {k!r} (saved {lay.get(k)!r}, this optimiser {mine[k]!r}); the flat "
                                     "buffers cannot be copied element for element -- reload through training/checkpoint.py")
        self.flat_m.copy_(sd["flat_m"])
        self.flat_v.copy_(sd["flat_v"])
        self.state.copy_(sd["state"])
