"""Per-kind throughput of k_wgrad_pairs: one job of each kind alone, 256 slices, n = 131 072 rows, LDS-DMA form against the register form."""
import torch
from holoscene_amd.hashencoder.backend import _backend as be

n, dev, bf = 131072, "cuda", torch.bfloat16
M = be.tp_rows(n)
g = torch.Generator().manual_seed(0)
tp = lambda w=256: (torch.randn(M * w, generator=g) * 0.1).to(dev).to(bf)  # noqa: E731
rm = lambda w: (torch.randn(n, w, generator=g) * 0.1).to(dev).to(bf)  # noqa: E731
A, B, A2, B2 = tp(), tp(), tp(), tp()
X80, X80b, G32, G32b, T128 = rm(80), rm(80), rm(32), rm(32), tp(128)
cases = {
    "256x256 tp 2 pairs": (((256, 256, "colsum"), 256, (A, B), (A2, B2)), 4 * M * 512),
    "256x256 tp 1 pair": (((256, 256), 256, (A, B), None), 2 * M * 512),
    "256x80 2 pairs": (((256, 80, "colsum"), 256, (A, X80), (A2, X80b)), 2 * (M * 512 + n * 160)),
    "32x256 2 pairs": (((32, 256), 256, (G32, B), (G32b, B2)), 2 * (M * 512 + n * 64)),
    "256x128 tp 1 pair": (((256, 128, "tp", "colsum"), 256, (A, T128), None), M * 512 + M * 256),
}
for name, (job, nbytes) in cases.items():
    for tag in ((), ("reg",), ("consecutive",)):
        j = (job[0] + tag,) + job[1:]
        for _ in range(3):
            be.wgrad_pairs([j], n)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            be.wgrad_pairs([j], n)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1000
        print(f"{name:22s} {'+'.join(tag) or 'dma':12s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
