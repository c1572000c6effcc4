"""k_wgrad_pairs on operands that are NOT in the 256 MB memory-side cache: four operand sets (1 GB) visited in turn.  The trunk backward's
three jobs at the bench's size (100 352 samples), slices cut as the model cuts them; LDS-DMA form against the register form."""
import sys
import torch
from holoscene_amd.hashencoder.backend import _backend as be
from holoscene_amd.model.network import _pair_slices

n, dev, bf = 100352, "cuda", torch.bfloat16
M = be.tp_rows(n)
T = M // 32
tp = lambda w=256: torch.randn(M * w, device=dev).mul_(0.1).to(bf)  # noqa: E731
rm = lambda w: torch.randn(n, w, device=dev).mul_(0.1).to(bf)  # noqa: E731
sets = [dict(A1=tp(), H0=tp(), V1=tp(), U0b=tp(), A0=tp(), V0=tp(), Xp=rm(80), UXb=rm(80), gy=rm(32), oh=rm(32), H1=tp(), U1b=tp()) for _ in range(3)]
s1, s0, s2 = _pair_slices([((256, 256), T, 2, True), ((256, 80), T, 2, True), ((32, 256), T, 2, True)])
print("slices", s1, s0, s2)


def jobs(s, tag, which):
    all3 = [((256, 256, "colsum") + tag, s1, (s["A1"], s["H0"]), (s["V1"], s["U0b"])), ((256, 80, "colsum") + tag, s0, (s["A0"], s["Xp"]), (s["V0"], s["UXb"])),
            ((32, 256) + tag, s2, (s["gy"], s["H1"]), (s["oh"], s["U1b"]))]
    if which == "all":
        return all3
    j = all3[int(which)]
    return [(j[0], 256) + j[2:]]


for which in ("all", "0", "1", "2"):
    for tag in ((), ("reg",), ("consecutive",)):
        for i in range(3):
            be.wgrad_pairs(jobs(sets[i], tag, which), n)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(21):
            be.wgrad_pairs(jobs(sets[i % 3], tag, which), n)
        b.record()
        torch.cuda.synchronize()
        print(which, "+".join(tag) or "dma", round(a.elapsed_time(b) / 21 * 1000, 1), "us (host clock)")
