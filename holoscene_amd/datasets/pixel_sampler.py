"""The class-balanced pixel draw of one training batch (reference: NSDataset.__getitem__, datasets/ns_dataset.py:409-430).

The reference draws a dozen ``torch.randperm`` per batch in 8 DataLoader worker processes (training/holoscene_train.py:124-129): ~4.5 ms
of host time per batch at 512 x 512 pixels and 32 classes, twice a whole training iteration here.  On a CUDA device the draw is ONE
launch on the training stream instead (csrc/batch_ops.hip: hs_draw_pixels) -- per class a uniformly random subset of its pixel list,
then the uniform half -- written into a static index tensor that the batch gather reads through a cached launch plan.  No host
threads, no ring, no host->device copies; (seed, counter) names a batch, so two samplers with one seed serve one sequence.  A new
draw per iteration, as the reference has it -- nothing is ever served twice.

On a CPU device (tests) the same rule runs on the host with a torch.Generator.  The reference's own permutations stay injectable
(`host_indices(frame, draws=...)`) for the parity fixtures.
"""
import collections

import torch


class PixelSampler:
    def __init__(self, class_pixels, total_pixels, num_pixels, device, seed=0):
        """class_pixels: per frame, the list of int64 host tensors `torch.nonzero(semantic == class)` for the classes present in that
        frame, background first (ns_dataset.py:419-421)."""
        self.device = torch.device(device)
        self.R, self.total = int(num_pixels), int(total_pixels)
        self._host = class_pixels
        self._gen = torch.Generator().manual_seed(int(seed))
        self._seed, self._counter = int(seed), 0
        self.half = self.R // 2
        self._frames = None
        self.idx = torch.zeros(self.R, dtype=torch.int64, device=self.device)      # static: the gather plans point at it

    def quotas(self, frame):
        n_cls = len(self._host[frame])
        per_class = self.half // n_cls
        return n_cls, per_class, self.half - per_class * (n_cls - 1)

    def count(self, frame):
        """Rays the rule yields for `frame` (< num_pixels when a class has fewer pixels than its quota, ns_dataset.py:422-427)."""
        n_cls, per_class, n_bg = self.quotas(frame)
        return sum(min(len(p), n_bg if i == 0 else per_class) for i, p in enumerate(self._host[frame])) + (self.R - self.half)

    # ------------------------------------------------------------------ host rule (CPU device, injected draws)
    def host_indices(self, frame, draws=None, gen=None):
        """Pixel indices of one batch of `frame` (host tensor, int64).  draws: optional iterator over the permutations ``torch.randperm``
        returned in the reference, in call order (one per class that has more pixels than its quota, then the uniform one)."""
        gen = self._gen if gen is None else gen
        n_cls, per_class, n_bg = self.quotas(frame)
        it = iter(draws) if draws is not None else None

        def perm(n):
            if it is not None:
                p = torch.as_tensor(next(it)).long()
                assert p.numel() == n, "injected permutation has the wrong length"
                return p
            return torch.randperm(n, generator=gen)

        chosen = []
        for i, pix in enumerate(self._host[frame]):
            want = n_bg if i == 0 else per_class
            if len(pix) > want:
                pix = pix[perm(len(pix))[:want]]
            chosen.append(pix)
        chosen.append(perm(self.total)[: self.R - self.half])
        return torch.cat(chosen)

    # ------------------------------------------------------------------ device rule
    def _device_frames(self):
        if self._frames is None:
            self._frames = []
            for f, lists in enumerate(self._host):
                n_cls, per_class, n_bg = self.quotas(f)
                sizes = [len(p) for p in lists]
                ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.int32)
                takes = [min(s, n_bg if i == 0 else per_class) for i, s in enumerate(sizes)] + [self.R - self.half]
                off = torch.tensor([0] + list(torch.tensor(takes).cumsum(0).tolist()), dtype=torch.int32)
                pix = torch.cat([p.reshape(-1) for p in lists]).to(torch.int32)
                self._frames.append((ptr.to(self.device), pix.to(self.device), off.to(self.device), int(off[-1])))
        return self._frames

    def skip(self):
        """Advance the stream by one batch without drawing it (a batch the caller refuses keeps the sequence of the others)."""
        self._counter += 1

    def draw(self, frame, gather=None):
        """The next batch of `frame` -> (static int64 index tensor, number of valid entries).  On CUDA the indices are written by one
        launch on the current stream (the previous batch's readers are ahead of it in stream order); gather: a backend.gather_plan() whose
        jobs are indexed by self.idx -- the rows of the batch are then gathered by the same launch (hs_draw_gather)."""
        if self.device.type != "cuda":
            idx = self.host_indices(frame)
            self.idx[:idx.numel()] = idx
            return self.idx, int(idx.numel())
        from ..hashencoder import backend as _be
        ptr, pix, off, n = self._device_frames()[frame]
        n_cls, per_class, n_bg = self.quotas(frame)
        self._counter += 1
        _be._backend.draw_pixels(ptr, pix, off, n_cls, per_class, n_bg, self.R - self.half, self.total, self._seed, self._counter, self.idx,
                                 n_out=n, gather=gather)
        return self.idx, n


class DeviceSchedule:
    """What the batch draw needs when it runs as a NODE OF THE TRAINING GRAPH (csrc/batch_ops.hip: hs_draw_gather_sched).  A launch between two graph
    replays leaves the chip idle around it (~14 us per iteration at configs[1], profiles/r06); inside the graph its arguments cannot come from the
    host, so they live on the device: a ring `sched` of the frames of the next batches and the batch number `cursor`, which the launch itself advances.
    ONE schedule per dataset, shared by the launch plans of all its destination blocks (the graph variants of a trainer).
    Batch number b = the sampler's counter before the draw, the draw's own counter b + 1: a batch is the same batch whichever path draws it, and the
    frames come from ONE queue (`frames`) that the eager paths consume too -- interleaving them walks one sequence.

    Host protocol around a replay of a graph that contains a ScheduledDraw.launch(): before_replay() (ring covers the batch, cursor holds its number: a
    host->device copy only every `n_sched` batches, or after an eager draw moved the counter), after_replay() (bookkeeping).  resync(): the device cursor
    moved without the host's bookkeeping (the warm-up passes of a capture)."""

    def __init__(self, sampler, frames, n_sched=512):
        self.sampler, self.frames, self.n = sampler, frames, int(n_sched)
        dev = sampler.device
        self.sched = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.cursor = torch.zeros(2, dtype=torch.int64, device=dev)
        self.lo = self.hi = 0           # the ring holds the frames of batches [lo, hi)
        self.cursor_at = None           # the batch number the device cursor is known to hold

    def ensure_ring(self, b, ahead=0):
        """The ring holds the frames of batches b .. b + ahead (refilled for b .. b + n - 1 otherwise)."""
        if not (self.lo <= b and b + ahead < self.hi):
            q = self.frames.queue
            while len(q) < self.n:
                q.append(self.frames.pick())
            ring = [0] * self.n
            for i in range(self.n):
                ring[(b + i) % self.n] = q[i]
            self.sched.copy_(torch.tensor(ring, dtype=torch.int32))     # (stream-ordered behind every replay that read the old content)
            self.lo, self.hi = b, b + self.n

    def before_replay(self):
        b = self.sampler._counter
        self.ensure_ring(b)
        self.set_cursor(b)

    def after_replay(self):
        self.frames.take()
        self.sampler._counter += 1
        self.cursor_at += 1

    def resync(self):
        self.cursor_at = None

    def set_cursor(self, b):
        if self.cursor_at != b:
            self.cursor.copy_(torch.tensor([b, 0], dtype=torch.int64))
            self.cursor_at = b


class ScheduledDraw:
    """The draw + gather of one destination block as a launch driven by a DeviceSchedule.  jobs: (src | per-frame list of src, dst, idx | None)."""

    def __init__(self, dataset, schedule, jobs):
        from ..hashencoder import backend as _be
        self.dataset, self.schedule = dataset, schedule
        sampler = schedule.sampler
        descs = []
        for f, (ptr, pix, off, n) in enumerate(sampler._device_frames()):
            n_cls, per_class, n_bg = sampler.quotas(f)
            descs.append((ptr, pix, off, n_cls, per_class, n_bg))
        self.plan = _be._backend.draw_sched_plan(descs, jobs, sampler.idx, schedule.sched, schedule.cursor, sampler._seed, 1)
        self.before_replay, self.after_replay = schedule.before_replay, schedule.after_replay
        self.holds = None       # draw-ahead protocol: the batch number the destination block holds (None: unknown)

    def resync(self):
        self.schedule.resync()
        self.holds = None

    # ---- draw-ahead: iteration b's graph draws batch b + 1 into the block late in its backward pass (hs_hash_bwd_draw), off the next iteration's
    # critical path.  The block must hold batch b when the replay starts: it does if the previous replay drew it, else it is drawn here, eagerly.
    def before_replay_ahead(self):
        sc = self.schedule
        b = sc.sampler._counter
        sc.ensure_ring(b, ahead=1)
        if self.holds != b or sc.cursor_at != b + 1:
            sc.set_cursor(b)
            self.launch()                   # batch b into the block, eagerly; the launch leaves the cursor at b + 1
            sc.cursor_at = b + 1
            self.holds = b

    def after_replay_ahead(self):
        sc = self.schedule
        sc.frames.take()
        sc.sampler._counter += 1
        sc.cursor_at += 1
        self.holds = sc.sampler._counter

    def launch(self):
        from ..hashencoder import backend as _be
        _be._backend.draw_gather_sched(*self.args())

    def args(self):
        """(plan, n_uniform, total_pixels, n_out): what hs_draw_gather_sched takes -- or hs_iter_prologue_draw, which takes this draw along
        (backend.iter_prologue(draw=...))."""
        s = self.schedule.sampler
        return self.plan, s.R - s.half, s.total, s.R


class FrameQueue:
    """The frames of the next batches in the order every path takes them: `take()` = the frame of the next batch (picked now unless a
    ScheduledDraw already picked ahead to fill its device ring)."""

    def __init__(self, pick):
        self.pick, self.queue = pick, collections.deque()

    def take(self):
        return self.queue.popleft() if self.queue else self.pick()

    def untake(self, frame):
        self.queue.appendleft(frame)
