"""Ring of pre-drawn (frame, pixel-index) batches that is REFILLED behind the consumer.

The reference draws a new frame and new ``torch.randperm`` pixel subsets for every iteration in 8 DataLoader worker processes
(datasets/ns_dataset.py:380-430, training/holoscene_train.py:124-129) -- a batch costs ~4.5 ms of host time at 512 x 512 pixels and
32 classes (one 262 144-element permutation plus one per class), more than a whole training iteration takes on the GPU here.  The
mirror of those workers: `workers` host threads (torch's CPU kernels release the GIL), each owning every `workers`-th slot of the ring
with its own generator, so the sequence of batches is a deterministic function of (seed, workers) whatever the thread timing.

Per slot: a STATIC device index tensor (the training graph's batch gather reads it through a cached launch plan), a pinned staging
buffer, and two events -- `ready` (the slot's host->device copy, recorded on the ring's copy stream; the consumer's stream waits for it)
and `consumed` (recorded by the consumer after the launch that read the slot; the refilling thread waits for it on the HOST before it
touches the slot, which also bounds how far the host can run ahead of the GPU: one ring).  A slot is redrawn as soon as it has been
consumed, so no batch is ever served twice.

On a CPU device (tests) there are no threads: a slot is redrawn synchronously when it is released.
"""
import queue
import random
import threading
import weakref

import torch


class _Slot:
    __slots__ = ("i", "idx", "fidx", "pin_idx", "pin_f", "frame", "count", "ready", "consumed", "filled", "error", "serial")


class BatchRing:
    def __init__(self, draw, capacity, device, ring=64, workers=8, seed=0, redraw=True):
        """draw(gen: torch.Generator, py: random.Random) -> (frame: int, idx: int64 host tensor of at most `capacity` pixels).
        redraw=False: the first fill is replayed for ever (tests that overfit a fixed set of batches)."""
        self._redraw = bool(redraw)
        self.device = torch.device(device)
        self._draw, self._cap = draw, int(capacity)
        self._threaded = self.device.type == "cuda" and workers > 0
        self._n = int(ring)
        workers = max(1, min(int(workers), self._n)) if self._threaded else 1
        self._workers = workers
        self._gens = [(torch.Generator().manual_seed(seed * 1009 + w), random.Random(seed * 1009 + w)) for w in range(workers)]
        self._slots = []
        for i in range(self._n):
            s = _Slot()
            s.i, s.frame, s.count, s.error, s.serial = i, -1, 0, None, 0
            s.idx = torch.zeros(self._cap, dtype=torch.int64, device=self.device)
            s.fidx = torch.zeros(1, dtype=torch.int64, device=self.device)
            s.filled = threading.Event()
            if self._threaded:
                s.pin_idx = torch.zeros(self._cap, dtype=torch.int64).pin_memory()
                s.pin_f = torch.zeros(1, dtype=torch.int64).pin_memory()
                s.ready, s.consumed = torch.cuda.Event(), torch.cuda.Event()
            else:
                s.pin_idx = s.pin_f = s.ready = s.consumed = None
            self._slots.append(s)
        self._cursor = 0
        self._started = False
        self._stop = False
        self._queues, self._threads = [], []
        self._copy_stream = None

    # ------------------------------------------------------------------ producer side
    def _fill(self, s, w):
        gen, py = self._gens[w]
        frame, idx = self._draw(gen, py)
        n = int(idx.numel())
        if n > self._cap:
            raise RuntimeError(f"batch of {n} pixels exceeds the ring's capacity {self._cap}")
        if self._threaded:
            s.pin_idx[:n].copy_(idx)
            s.pin_f[0] = frame
            with torch.cuda.stream(self._copy_stream):
                s.idx[:n].copy_(s.pin_idx[:n], non_blocking=True)
                s.fidx.copy_(s.pin_f, non_blocking=True)
                s.ready.record(self._copy_stream)
        else:
            s.idx[:n].copy_(idx)
            s.fidx[0] = frame
        s.frame, s.count = int(frame), n
        s.serial += 1

    @staticmethod
    def _worker(ref, q, w, device):
        """Thread body.  Holds the ring through a weak reference only, so that dropping the dataset ends its threads (close())."""
        torch.cuda.set_device(device)
        while True:
            s = q.get()
            ring = ref()
            if s is None or ring is None or ring._stop:
                return
            try:
                if s.serial:                     # a slot that has been served before: wait (host side) until the launch that read it has
                    s.consumed.synchronize()     # run -- its index tensor and its staging buffer are then free to be overwritten
                ring._fill(s, w)
            except BaseException as e:           # noqa: BLE001 -- handed to the consumer, which re-raises
                s.error = e
            s.filled.set()
            del ring

    def _start(self):
        self._started = True
        if not self._threaded:
            for s in self._slots:
                self._fill(s, 0)
                s.filled.set()
            return
        self._copy_stream = torch.cuda.Stream(self.device)
        self._queues = [queue.SimpleQueue() for _ in range(self._workers)]
        for s in self._slots:
            self._queues[s.i % self._workers].put(s)
        for w in range(self._workers):
            t = threading.Thread(target=BatchRing._worker, args=(weakref.ref(self), self._queues[w], w, self.device), daemon=True,
                                 name=f"hs-batch-ring-{w}")
            t.start()
            self._threads.append(t)

    # ------------------------------------------------------------------ consumer side
    def acquire(self):
        """The next slot, filled and safe to read on the current stream.  Pair with release()."""
        if not self._started:
            self._start()
        s = self._slots[self._cursor % self._n]
        self._cursor += 1
        s.filled.wait()
        if s.error is not None:
            raise s.error
        if self._threaded:
            torch.cuda.current_stream(self.device).wait_event(s.ready)
        return s

    def release(self, s):
        """The launches reading slot s have been enqueued on the current stream: hand it back to be redrawn."""
        if not self._redraw:
            return
        s.filled.clear()
        if self._threaded:
            s.consumed.record(torch.cuda.current_stream(self.device))
            self._queues[s.i % self._workers].put(s)
        else:
            self._fill(s, 0)
            s.filled.set()

    def close(self):
        self._stop = True
        for q in self._queues:
            q.put(None)
        me = threading.current_thread()
        for t in self._threads:
            if t is not me:
                t.join(timeout=5.0)
        self._threads = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
