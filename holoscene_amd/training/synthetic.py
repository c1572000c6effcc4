"""Synthetic Stage-1 inputs (no dataset, no network in this environment) -- SURVEY.md section 8(d).

Frames: H x W image, pinhole K (fx=fy=W/2, cx=cy=W/2 for the square case), F camera-to-world poses on a
circle of radius 0.7 in the z=0 plane looking at the origin; instance mask = K vertical stripes;
rgb ~ U[0,1], depth ~ U[0.1,1], unit normals from N(0,I), mask = 1.  Pixel batches follow the
reference's class-balanced rule (datasets/ns_dataset.py:409-430): half of the rays split evenly over
the classes present (background takes the remainder), the other half uniform over the image.
The frames are generated once and kept resident in HBM; every iteration draws a new frame and new random
pixel subsets, as ns_dataset.py:380-430 has it -- by one launch on the training stream
(datasets/pixel_sampler.py, csrc/batch_ops.hip: hs_draw_pixels; the reference's 8 DataLoader workers spend
~4.5 ms of host time per batch on it).
"""
import random

import numpy as np
import torch

from ..datasets.pixel_sampler import DeviceSchedule, FrameQueue, PixelSampler, ScheduledDraw


def look_at_pose(eye, target=(0.0, 0.0, 0.0)):
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    P = np.eye(4)
    P[:3, 0], P[:3, 1], P[:3, 2], P[:3, 3] = right, down, fwd, eye
    return torch.from_numpy(P).float()


class SyntheticScene:
    def __init__(self, num_rays, num_classes, img_res=(512, 512), num_frames=8, ring=64, seed=1234, device="cuda", redraw=True):
        """ring / redraw=False: tests that overfit a FIXED set of `ring` batches (drawn once, replayed for ever)."""
        self.R, self.K = num_rays, num_classes
        self.H, self.W = img_res
        self.F = num_frames
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        self.rng = np.random.RandomState(seed)
        intr = torch.eye(4)
        intr[0, 0] = intr[1, 1] = self.W / 2
        intr[0, 2], intr[1, 2] = self.W / 2, self.H / 2
        self.intrinsics = intr[None].to(self.device)
        ang = np.linspace(0, 2 * np.pi, num_frames, endpoint=False)
        self.poses = torch.stack([look_at_pose((0.7 * np.cos(a), 0.7 * np.sin(a), 0.0)) for a in ang]).to(self.device)
        npix = self.H * self.W
        cols = torch.arange(self.W) * num_classes // self.W                      # stripe id per column
        self.segs = cols[None, :].expand(self.H, self.W).reshape(npix, 1).contiguous().to(self.device)
        self.rgb = torch.rand(num_frames, npix, 3, generator=g).to(self.device)
        self.depth = (torch.rand(num_frames, npix, 1, generator=g) * 0.9 + 0.1).to(self.device)
        self.normal = torch.nn.functional.normalize(torch.randn(num_frames, npix, 3, generator=g), dim=-1).to(self.device)
        ys, xs = torch.meshgrid(torch.arange(self.H), torch.arange(self.W), indexing="ij")
        self.uv_all = torch.stack([xs, ys], -1).reshape(npix, 2).float().to(self.device)
        self._class_pixels = [torch.nonzero(self.segs.cpu().reshape(-1) == c).reshape(-1) for c in range(num_classes)]
        self._sampler = PixelSampler([self._class_pixels] * num_frames, npix, num_rays, self.device, seed=seed)
        self._py = random.Random(seed)
        self._frames = FrameQueue(lambda: self._py.randint(0, self.F - 1))      # ns_dataset.py:383
        self._plans, self._const_done, self._schedule = {}, set(), None
        self._fidx = torch.arange(num_frames, dtype=torch.int64, device=self.device)
        self._fixed, self._cursor = None, 0
        if not redraw:      # a fixed set of batches, drawn once on the host
            g2 = torch.Generator().manual_seed(seed)
            self._fixed = []
            for _ in range(ring):
                f = self._py.randint(0, self.F - 1)
                self._fixed.append((f, self._sampler.host_indices(f, gen=g2).to(self.device)))

    def _next(self):
        """(frame, static index tensor) of the next batch."""
        if self._fixed is not None:
            f, idx = self._fixed[self._cursor % len(self._fixed)]
            self._cursor += 1
            self._sampler.idx.copy_(idx)
            return f, self._sampler.idx
        f = self._frames.take()
        idx, _ = self._sampler.draw(f)
        return f, idx

    def next_batch(self):
        frame, idx = self._next()
        model_input = {"uv": self.uv_all[idx][None], "intrinsics": self.intrinsics, "pose": self.poses[frame][None]}
        gt = {"rgb": self.rgb[frame][idx][None], "depth": self.depth[frame][idx][None], "normal": self.normal[frame][idx][None],
              "mask": torch.ones(1, idx.numel(), 1, device=self.device), "segs": self.segs[idx][None]}
        return torch.tensor([frame]), model_input, gt

    def write_batch(self, dst_input, dst_gt):
        """next_batch() written straight into existing buffers (the training graph's static input block): ONE launch draws the pixel
        indices and gathers their rows (csrc/batch_ops.hip: hs_draw_gather) -- instead of six indexing launches plus the copies into
        the block.  Same generator state, same counter: interleaving next_batch() and write_batch() walks the same
        sequence of batches.  The launch plan of a frame holds pointers only -- the index tensor is static, its content is redrawn."""
        from ..hashencoder import backend as _be
        fused = self._fixed is None
        if fused:
            frame, idx = self._frames.take(), self._sampler.idx     # the draw itself rides in the gather launch below
        else:
            frame, idx = self._next()
        tag = dst_input["uv"].data_ptr()
        key = (frame, tag)
        plan = self._plans.get(key)
        if plan is None:
            fidx = self._fidx[frame:frame + 1]
            plan = self._plans[key] = _be._backend.gather_plan([
                (self.uv_all, dst_input["uv"], idx), (self.poses, dst_input["pose"], fidx), (self.rgb[frame], dst_gt["rgb"], idx),
                (self.depth[frame], dst_gt["depth"], idx), (self.normal[frame], dst_gt["normal"], idx), (self.segs, dst_gt["segs"], idx)])
        self._write_constants(dst_input, dst_gt)
        if fused:
            self._sampler.draw(frame, gather=plan)      # pixel draw + row gather: one launch (csrc/batch_ops.hip: hs_draw_gather)
        else:
            _be._backend.gather_rows(plan)

    def _write_constants(self, dst_input, dst_gt):
        tag = dst_input["uv"].data_ptr()
        if tag not in self._const_done:     # per-batch constants of this scene: written once per destination block
            dst_input["intrinsics"].copy_(self.intrinsics)
            dst_gt["mask"].fill_(1.0)
            self._const_done.add(tag)

    def peek_batch(self):
        """next_batch() without consuming it: the next draw -- by any path -- yields the same batch again."""
        if self._fixed is not None:
            cur = self._cursor
            out = self.next_batch()
            self._cursor = cur
            return out
        c, g = self._sampler._counter, self._sampler._gen.get_state()      # (the host rule of a CPU scene draws from the generator, the device rule from the counter)
        out = self.next_batch()
        self._frames.untake(int(out[0][0]))
        self._sampler._counter = c
        self._sampler._gen.set_state(g)
        return out

    def scheduled_draw(self, dst_input, dst_gt):
        """The draw + gather of write_batch() as a launch whose arguments live on the device (datasets/pixel_sampler.py: ScheduledDraw), or None
        when this scene cannot (fixed batches; a frame whose rule yields fewer rays than the block holds)."""
        if self._fixed is not None or self.device.type != "cuda":
            return None
        if any(self._sampler.count(f) != dst_input["uv"].shape[1] for f in range(self.F)):
            return None
        self._write_constants(dst_input, dst_gt)
        idx = self._sampler.idx
        per = lambda t: [t[f] for f in range(self.F)]  # noqa: E731
        jobs = [(self.uv_all, dst_input["uv"], idx), (self.poses, dst_input["pose"], None), (per(self.rgb), dst_gt["rgb"], idx),
                (per(self.depth), dst_gt["depth"], idx), (per(self.normal), dst_gt["normal"], idx), (self.segs, dst_gt["segs"], idx)]
        if self._schedule is None:
            self._schedule = DeviceSchedule(self._sampler, self._frames)
        return ScheduledDraw(self, self._schedule, jobs)
