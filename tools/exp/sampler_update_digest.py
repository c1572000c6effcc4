"""Exactness + timing harness for k_sampler_update: prints a digest of (beta, max beta, merged depths) on five shapes and the kernel
time.  A change that is meant to leave the arithmetic alone must print the same digests before and after."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holoscene_amd.hashencoder import backend  # noqa: E402

be = backend._backend
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
digests = []
for m_old, s_new, R in ((128, 64, 1024), (192, 64, 1024), (384, 64, 1024), (448, 64, 1000), (640, 128, 512)):
    ld = m_old + s_new
    z_old = torch.sort(torch.rand(R, m_old, generator=g) * 3.0, dim=1)[0]
    z_new = torch.sort(torch.rand(R, s_new, generator=g) * 3.0, dim=1)[0]
    centre = torch.rand(R, 1, generator=g) * 2.0 + 0.5
    rad = torch.rand(R, 1, generator=g) * 0.4 + 0.05
    miss = (torch.rand(R, 1, generator=torch.Generator().manual_seed(3)) - 0.3).clamp(min=0)

    def sdf_of(z):      # a sphere crossing along the ray, sometimes missed
        return (z - centre).abs() - rad + miss
    z = torch.zeros(R, ld)
    z[:, :m_old] = z_old
    sdf = torch.zeros(R, ld)
    sdf[:, :m_old] = sdf_of(z_old)
    z, sdf = z.to(dev), sdf.to(dev)
    samples, new_sdf = z_new.to(dev).contiguous(), sdf_of(z_new).to(dev).contiguous()
    beta = (torch.rand(R, generator=g) * 0.5 + 0.05).to(dev)
    beta0 = torch.tensor([0.01], device=dev)
    bmax = torch.zeros(1, device=dev)
    z0, s0, b0 = z.clone(), sdf.clone(), beta.clone()
    be.sampler_update(z, sdf, m_old, samples, new_sdf, beta, beta0, 0.1, 10, bmax)
    torch.cuda.synchronize()
    digests.append(hashlib.sha1(beta.cpu().numpy().tobytes() + bmax.cpu().numpy().tobytes() + z.cpu().numpy().tobytes()).hexdigest()[:12])
    frac = float((beta > 0.01).float().mean())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for it in range(25):
        z.copy_(z0); sdf.copy_(s0); beta.copy_(b0)
        ev[0].record()
        be.sampler_update(z, sdf, m_old, samples, new_sdf, beta, beta0, 0.1, 10, bmax)
        ev[1].record()
        torch.cuda.synchronize()
        if it >= 5:
            ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts.sort()
    print(f"m={ld:4d} R={R} searching rays {frac:.2f}  {ts[len(ts) // 2]:7.1f} us  digest {digests[-1]}", flush=True)
print("ALL", hashlib.sha1("".join(digests).encode()).hexdigest()[:16], "(reference: 237a2ab748fc7064)")
