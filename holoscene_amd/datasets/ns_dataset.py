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
static input block (``write_batch``; csrc/encode_ops.hip: hs_gather_rows).  Batches are drawn ahead into a ring, off the critical path,
as the reference's workers do.  Loading the image files stays with the reference's ``NSDataset`` (``from_reference``).
"""
import random

import numpy as np
import torch


class ResidentNSDataset:
    def __init__(self, rgb_images, depth_images, normal_images, mask_images, semantic_images, semantic_images_classes, intrinsics_all, pose_all,
                 img_res, num_pixels, fix_length=0, device="cuda", ring=64, seed=0):
        """*_images: per-frame flat tensors as NSDataset stores them ([H*W, C]; lists or stacked); semantic_images_classes: per frame the
        sorted class ids present in it (ns_dataset.py:307-308)."""
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
        self._gen = torch.Generator().manual_seed(seed)
        self._py = random.Random(seed)
        self._ring_len, self._ring, self._cursor = ring, [], 0
        self._plans, self._const_done = {}, set()

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
        half = self.sampling_size // 2
        n_cls = len(self.classes[frame])
        per_class = half // n_cls
        n_bg = half - per_class * (n_cls - 1)
        it = iter(draws) if draws is not None else None

        def perm(n):
            if it is not None:
                p = torch.as_tensor(next(it)).long()
                assert p.numel() == n, "injected permutation has the wrong length"
                return p
            return torch.randperm(n, generator=self._gen)

        chosen = []
        for i, pix in enumerate(self._class_pixels[frame]):
            want = n_bg if i == 0 else per_class
            if len(pix) > want:
                pix = pix[perm(len(pix))[:want]]
            chosen.append(pix)
        chosen.append(perm(self.total_pixels)[: self.sampling_size - half])
        return torch.cat(chosen)

    def pick_frame(self, idx=None):
        if self.fix_length != 0 or idx is None:      # :382-383
            return self._py.randint(0, self.n_images - 1)
        return int(idx)

    # ------------------------------------------------------------------ batches
    def get(self, frame, sampling_idx):
        """(indices, model_input, ground_truth) as ``collate_fn([__getitem__(.)])`` returns them (:393-455; batch dimension 1)."""
        idx = sampling_idx.to(self.device)
        sample = {"uv": self.uv_all[idx][None], "intrinsics": self.intrinsics_all[frame][None], "pose": self.pose_all[frame][None]}
        gt = {"rgb": self.rgb[frame][idx][None], "depth": self.depth[frame][idx][None], "mask": self.mask[frame][idx][None],
              "normal": self.normal[frame][idx][None], "segs": self.segs[frame][idx][None]}
        return torch.tensor([frame]), sample, gt

    def _fill_ring(self):
        while len(self._ring) < self._ring_len:
            f = self.pick_frame()
            idx = self.sample_indices(f)
            self._ring.append((f, idx.to(self.device), torch.tensor([f], dtype=torch.int64).to(self.device)))

    def next_batch(self):
        self._fill_ring()
        frame, idx, _ = self._ring[self._cursor % self._ring_len]
        self._cursor += 1
        return self.get(frame, idx)

    def write_batch(self, dst_input, dst_gt):
        """The next ring batch gathered straight into existing buffers (the training graph's static input block) by one launch."""
        from ..hashencoder import backend as _be
        self._fill_ring()
        slot = self._cursor % self._ring_len
        self._cursor += 1
        frame, idx, fidx = self._ring[slot]
        if idx.numel() != dst_input["uv"].shape[1]:
            raise RuntimeError(f"batch of {idx.numel()} rays (a class of frame {frame} has fewer pixels than its quota, ns_dataset.py:422-427) "
                               f"does not fit the static block of {dst_input['uv'].shape[1]}: use next_batch() / the eager path for such scenes")
        tag = dst_input["uv"].data_ptr()
        plan = self._plans.get((slot, tag))
        if plan is None:
            plan = self._plans[(slot, tag)] = _be._backend.gather_plan([
                (self.uv_all, dst_input["uv"], idx), (self.pose_all, dst_input["pose"], fidx), (self.intrinsics_all, dst_input["intrinsics"], fidx),
                (self.rgb[frame], dst_gt["rgb"], idx), (self.depth[frame], dst_gt["depth"], idx), (self.normal[frame], dst_gt["normal"], idx),
                (self.mask[frame], dst_gt["mask"], idx), (self.segs[frame], dst_gt["segs"], idx)])
        _be._backend.gather_rows(plan)
