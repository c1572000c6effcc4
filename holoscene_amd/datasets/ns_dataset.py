"""HBM-resident mirror of the reference's training-batch producer.

The reference keeps every frame of a scene in host RAM as flat per-pixel tensors (datasets/ns_dataset.py:246-318: rgb / depth / normal /
mask / instance images, intrinsics, poses) and produces one batch per iteration in DataLoader worker processes
(``NSDataset.__getitem__`` :380-455): a frame -- random when ``fix_length != 0`` (:383) --, then the class-balanced pixel subset of
:409-430: half of the rays split evenly over the instance classes present in THAT frame (class 0, the background, takes the
remainder; a class with fewer pixels than its quota contributes all of them), the other half uniform over the image.  The batch then
crosses the process boundary and the PCIe bus every iteration (holoscene_train.py:332-336).

``ResidentNSDataset`` holds the same tensors on the device, applies the same rule (``sample_indices``; every permutation injectable so
that the reference's own draws reproduce its batches index for index -- tests/golden/ns_sampler.npz) and hands a batch over either as
the dictionaries ``__getitem__`` + ``collate_fn`` produce (``next_batch``) or gathered by ONE launch straight into the training graph's
static input block (``write_batch``; csrc/encode_ops.hip: hs_gather_rows).  Every batch is a NEW draw -- a new frame and new random pixel
subsets per iteration, as ns_dataset.py:380-430 has it -- made by one launch on the training stream (datasets/pixel_sampler.py,
csrc/batch_ops.hip: hs_draw_pixels), where the reference spends ~4.5 ms of host time per batch in DataLoader workers.  Loading the image
files stays with the reference's ``NSDataset`` (``from_reference``).
"""
import random

import numpy as np
import torch

from .pixel_sampler import DeviceSchedule, FrameQueue, PixelSampler, ScheduledDraw


def rank_seed(seed, rank=None):
    """Data-parallel ranks must not draw the same frames and pixels (an effective batch of 1/N): the stream of a rank is derived from
    its rank: seed + 7919 rank, i.e. rank 0 keeps the bare seed (a single process and rank 0 of a job read the same batches).
    (Stage1Trainer reseeds the MODEL's draws with seed + 7919 (rank + 1) after the common initialisation; the two streams are unrelated.)
    rank None: the process group's rank when one is alive."""
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return int(seed) + 7919 * int(rank) if rank else int(seed)


class ResidentNSDataset:
    def __init__(self, rgb_images, depth_images, normal_images, mask_images, semantic_images, semantic_images_classes, intrinsics_all, pose_all,
                 img_res, num_pixels, fix_length=0, device="cuda", seed=0, rank=None):
        """*_images: per-frame flat tensors as NSDataset stores them ([H*W, C]; lists or stacked); semantic_images_classes: per frame the
        sorted class ids present in it (ns_dataset.py:307-308).  seed / rank: every data-parallel rank draws its own stream (rank_seed)."""
        dev = self.device = torch.device(device)
        stack = lambda t: (torch.stack(list(t)) if not torch.is_tensor(t) else t).float()  # noqa: E731
        self.rgb, self.depth, self.normal, self.mask = (stack(t).to(dev) for t in (rgb_images, depth_images, normal_images, mask_images))
        self.segs = stack(semantic_images).to(dev)
        self.classes = [torch.as_tensor(c).reshape(-1).tolist() for c in semantic_images_classes]
        self.intrinsics_all, self.pose_all = stack(intrinsics_all).to(dev), stack(pose_all).to(dev)
        self.img_res = tuple(int(v) for v in img_res)
        self.total_pixels = self.img_res[0] * self.img_res[1]
        self.n_images = self.rgb.shape[0]
        self.fix_length = fix_length
        self.sampling_size = num_pixels
        # uv of every pixel exactly as __getitem__ builds it (:390-392): mgrid over (H, W), flipped to (x, y)
        uv = np.mgrid[0:self.img_res[0], 0:self.img_res[1]].astype(np.int32)
        self.uv_all = torch.from_numpy(np.flip(uv, axis=0).copy()).float().reshape(2, -1).transpose(1, 0).contiguous().to(dev)
        # pixel lists per (frame, class): torch.nonzero(semantic_images[idx] == class) of :419-421, computed once
        segs_host = self.segs.reshape(self.n_images, -1).cpu()
        self._class_pixels = [[torch.nonzero(segs_host[f] == c).reshape(-1) for c in self.classes[f]] for f in range(self.n_images)]
        seed = rank_seed(seed, rank)
        self._gen = torch.Generator().manual_seed(seed)
        self._py = random.Random(seed)
        self._epoch = []                # fix_length == 0: frames of the current epoch still to be served (a shuffled DataLoader epoch)
        self._sampler = PixelSampler(self._class_pixels, self.total_pixels, num_pixels, dev, seed=seed)
        self._fidx = torch.arange(self.n_images, dtype=torch.int64, device=dev)      # per-frame index rows for write_batch's gather plans (no host->device copy in the loop)
        self._plans = {}
        self._frames = FrameQueue(lambda: self.pick_frame())     # the frames of the next batches, in the order every path takes them
        self._schedule = None

    @classmethod
    def from_reference(cls, ds, num_pixels, device="cuda", **kw):
        """ds: a constructed reference ``NSDataset`` (it has read the files)."""
        return cls(ds.rgb_images, ds.depth_images, ds.normal_images, ds.mask_images, ds.semantic_images, ds.semantic_images_classes,
                   ds.intrinsics_all, ds.pose_all, ds.img_res, num_pixels, fix_length=getattr(ds, "fix_length", 0), device=device, **kw)

    def __len__(self):
        return self.n_images if self.fix_length == 0 else self.fix_length

    # ------------------------------------------------------------------ the sampling rule
    def sample_indices(self, frame, draws=None):
        """Pixel indices of one batch of `frame` (host tensor, int64), ns_dataset.py:409-430.  draws: optional iterator over the
        permutations ``torch.randperm`` returned in call order (one per class that has more pixels than its quota, then the uniform one)."""
        return self._sampler.host_indices(frame, draws)

    def pick_frame(self, idx=None, py=None):
        """fix_length != 0: a random frame per item (:382-383).  fix_length == 0: the DataLoader hands out the frames of a shuffled
        epoch (holoscene_train.py:124-127, shuffle=True) -- every frame once before any repeats."""
        py = self._py if py is None else py
        if self.fix_length != 0:
            return py.randint(0, self.n_images - 1)
        if idx is not None:
            return int(idx)
        if not self._epoch:
            self._epoch = list(range(self.n_images))
            py.shuffle(self._epoch)
        return self._epoch.pop()

    # ------------------------------------------------------------------ batches
    def get(self, frame, sampling_idx):
        """(indices, model_input, ground_truth) as ``collate_fn([__getitem__(.)])`` returns them (:393-455; batch dimension 1)."""
        idx = sampling_idx.to(self.device)
        sample = {"uv": self.uv_all[idx][None], "intrinsics": self.intrinsics_all[frame][None], "pose": self.pose_all[frame][None]}
        gt = {"rgb": self.rgb[frame][idx][None], "depth": self.depth[frame][idx][None], "mask": self.mask[frame][idx][None],
              "normal": self.normal[frame][idx][None], "segs": self.segs[frame][idx][None]}
        return torch.tensor([frame]), sample, gt

    def next_batch(self):
        frame = self._frames.take()
        idx, n = self._sampler.draw(frame)
        return self.get(frame, idx[:n])

    def peek_batch(self):
        """next_batch() without consuming it: the next draw -- by any path -- yields the same batch again."""
        c, g = self._sampler._counter, self._sampler._gen.get_state()      # (host rule: the generator; device rule: the counter)
        out = self.next_batch()
        self._frames.untake(int(out[0][0]))
        self._sampler._counter = c
        self._sampler._gen.set_state(g)
        return out

    def scheduled_draw(self, dst_input, dst_gt):
        """The draw + gather of write_batch() as a launch whose arguments live on the device (pixel_sampler.py: DeviceSchedule / ScheduledDraw) --
        a node of the training graph -- or None when a frame's rule yields fewer rays than the block holds (ns_dataset.py:422-427)."""
        if self.device.type != "cuda" or any(self._sampler.count(f) != dst_input["uv"].shape[1] for f in range(self.n_images)):
            return None
        idx = self._sampler.idx
        per = lambda t: [t[f] for f in range(self.n_images)]  # noqa: E731
        jobs = [(self.uv_all, dst_input["uv"], idx), (self.pose_all, dst_input["pose"], None), (self.intrinsics_all, dst_input["intrinsics"], None),
                (per(self.rgb), dst_gt["rgb"], idx), (per(self.depth), dst_gt["depth"], idx), (per(self.normal), dst_gt["normal"], idx),
                (per(self.mask), dst_gt["mask"], idx), (per(self.segs), dst_gt["segs"], idx)]
        if self._schedule is None:
            self._schedule = DeviceSchedule(self._sampler, self._frames)
        return ScheduledDraw(self, self._schedule, jobs)

    def write_batch(self, dst_input, dst_gt):
        """The next batch gathered straight into existing buffers (the training graph's static input block): one launch for the draw
        and the gather together."""
        from ..hashencoder import backend as _be
        frame = self._frames.take()
        n = self._sampler.count(frame)
        if n != dst_input["uv"].shape[1]:
            self._sampler.skip()
            raise RuntimeError(f"batch of {n} rays (a class of frame {frame} has fewer pixels than its quota, ns_dataset.py:422-427) "
                               f"does not fit the static block of {dst_input['uv'].shape[1]}: use next_batch() / the eager path for such scenes")
        idx = self._sampler.idx
        # the launch plan holds pointers only: the index tensor is static (its CONTENT is redrawn), the image sources depend on the frame
        key = (frame, dst_input["uv"].data_ptr())
        plan = self._plans.get(key)
        if plan is None:
            fidx = self._fidx[frame:frame + 1]
            plan = self._plans[key] = _be._backend.gather_plan([
                (self.uv_all, dst_input["uv"], idx), (self.pose_all, dst_input["pose"], fidx), (self.intrinsics_all, dst_input["intrinsics"], fidx),
                (self.rgb[frame], dst_gt["rgb"], idx), (self.depth[frame], dst_gt["depth"], idx), (self.normal[frame], dst_gt["normal"], idx),
                (self.mask[frame], dst_gt["mask"], idx), (self.segs[frame], dst_gt["segs"], idx)])
        if self.device.type == "cuda":
            self._sampler.draw(frame, gather=plan)      # pixel draw + row gather: one launch (csrc/batch_ops.hip: hs_draw_gather)
        else:
            self._sampler.draw(frame)
            _be._backend.gather_rows(plan)
