"""tools/exp/appear2_check.py -- k_appear2_fwd (wave-tile colour branch) against k_appear_fwd on the same inputs: rgb, the saved layer
outputs (tile-packed decoded vs row-major), timing of both."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from holoscene_amd.hashencoder.backend import _backend as be
import rr_reference as R

dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(3)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731


def run(n):
    featc = rn(16, n, 2, sc=0.3)
    points, dirs, normals = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev), torch.nn.functional.normalize(rn(n, 3), dim=-1), torch.nn.functional.normalize(rn(n, 3), dim=-1)
    wc0, wc1, wr0, wr1, wr2 = rn(256, 32, sc=0.2), rn(256, 256, sc=0.07), rn(256, 337, sc=0.06), rn(256, 256, sc=0.07), rn(3, 256, sc=0.1)
    bs = (rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(3, sc=0.1))
    new = lambda r, c: torch.empty(r, c, device=dev, dtype=bf)  # noqa: E731
    W = {"Wc0": new(256, 32), "Wc1": new(256, 256), "Wr0f": new(256, 256), "Wr0p": new(256, 96), "Wr1": new(256, 256), "Wr2": new(32, 256)}
    be.pack_bf16([(wc0, W["Wc0"], 0, 0, 256, 32, False), (wc1, W["Wc1"], 0, 0, 256, 256, False), (wr0, W["Wr0f"], 0, 81, 256, 256, False),
                  (wr0, W["Wr0p"], 0, 0, 256, 81, False), (wr1, W["Wr1"], 0, 0, 256, 256, False), (wr2, W["Wr2"], 0, 0, 3, 256, False)])
    xin, hc, fv, r0, r1 = new(n, 128), new(n, 256), new(n, 256), new(n, 256), new(n, 256)
    rgb = torch.empty(n, 3, device=dev)
    old = lambda: be.appearance_fwd(featc, points, dirs, normals, W, bs, xin, hc, fv, r0, r1, rgb, None)  # noqa: E731
    old()
    P = be.appearance2_pack(wc0, wc1, wr0, wr1, wr2, bs)
    tiles = (n + 31) // 32
    tp = lambda ks: torch.zeros(tiles * ks * 64 * 8, device=dev, dtype=bf)  # noqa: E731
    XAt, HCt, FVt, R0t, R1t = tp(8), tp(16), tp(16), tp(16), tp(16)
    masks = torch.zeros(tiles * 3 * 64 * 4, device=dev, dtype=torch.int32)
    rgb2 = torch.empty(n, 3, device=dev)
    new_ = lambda: be.appearance2_fwd(featc, points, dirs, normals, P, XAt, HCt, FVt, R0t, R1t, masks, rgb2)  # noqa: E731
    new_()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))  # noqa: E731
    print(f"n={n}: rgb max abs diff {float((rgb2 - rgb).abs().max()):.3e}")
    for name, T, ref in (("hc", HCt, hc), ("fv", FVt, fv), ("r0", R0t, r0), ("r1", R1t, r1)):
        d = R.tp_decode(T, n)
        print(f"   {name}: relL2 {rel(d, ref.float()):.3e}  max abs {float((d - ref.float()).abs().max()):.3e}")
    # ReLU masks: word m[nd >> 1] of (tile, layer, lane): bit 8 (nd & 1) + (reg >> 1) + 16 (reg & 1) SET = unit off, for accumulator register reg of tile nd
    for layer, T in ((0, HCt), (1, R0t), (2, R1t)):
        d = R.tp_decode(T, tiles * 32) > 0                                   # [rows, 256]
        mk = masks.view(tiles, 3, 2, 32, 4)[:, layer]                        # [tile, h, row, word]
        bits = ((mk.unsqueeze(-1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1).bool()    # [tile, h, row, word, bit]
        got = torch.zeros(tiles, 32, 256, dtype=torch.bool, device=dev)
        for nd in range(8):
            for reg in range(16):
                for h in range(2):
                    got[:, :, 32 * nd + 8 * (reg >> 2) + 4 * h + (reg & 3)] = ~bits[:, h, :, nd >> 1, 8 * (nd & 1) + (reg >> 1) + 16 * (reg & 1)]
        print(f"   mask layer {layer}: mismatching bits {int((got.view(-1, 256)[:n] != d[:n]).sum())} of {n * 256}")

    def timed(fn, k=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(k):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / k * 1e3
    print(f"   k_appear_fwd {timed(old):.1f} us   k_appear2_fwd {timed(new_):.1f} us")
    # ---- backward: old kernel on its own saved activations (masks from r1, r0, hc) vs the wave-tile kernel on the forward's masks
    Wt = {"Wr2t": new(256, 32), "Wr1t": new(256, 256), "Wr0ft": new(256, 256), "Wr0nt": new(32, 256), "Wc1t": new(256, 256), "Wc0t": new(32, 256)}
    be.pack_bf16([(wr2, Wt["Wr2t"], 0, 0, 256, 3, True), (wr1, Wt["Wr1t"], 0, 0, 256, 256, True), (wr0, Wt["Wr0ft"], 0, 81, 256, 256, True),
                  (wr0, Wt["Wr0nt"], 0, 54, 27, 256, True), (wc1, Wt["Wc1t"], 0, 0, 256, 256, True), (wc0, Wt["Wc0t"], 0, 0, 32, 256, True)])
    g_rgb = rn(n, 3)
    gy, gA_r1, gA_r0, g_fv, gA_hc = new(n, 32), new(n, 256), new(n, 256), new(n, 256), new(n, 256)
    d_n, g_fc, gb = torch.empty(n, 3, device=dev), torch.empty(16, n, 2, device=dev), torch.zeros(5, 256, device=dev)
    oldb = lambda: be.appearance_bwd(g_rgb, rgb, normals, r1, r0, hc, Wt, gy, gA_r1, gA_r0, g_fv, gA_hc, d_n, g_fc, gb, None)  # noqa: E731
    oldb()
    sT = P["streamT"]
    gy2 = new(n, 32)
    GR1, GR0, GFV, GHC = tp(16), tp(16), tp(16), tp(16)
    d_n2, g_fc2, gb2 = torch.empty(n, 3, device=dev), torch.empty(16, n, 2, device=dev), torch.zeros(tiles, 4, device=dev)
    newb = lambda: be.appearance2_bwd(g_rgb, rgb2, normals, masks, sT, gy2, GR1, GR0, GFV, GHC, d_n2, g_fc2, gb2)  # noqa: E731
    newb()
    torch.cuda.synchronize()
    print(f"   bwd gy relL2 {rel(gy2.float(), gy.float()):.3e}  d_normals {rel(d_n2, d_n):.3e}  g_featc {rel(g_fc2, g_fc):.3e}  gb2 {rel(gb2.sum(0)[:3], gb[4, :3]):.3e}")
    for name, T, ref in (("gA_r1", GR1, gA_r1), ("gA_r0", GR0, gA_r0), ("g_fv", GFV, g_fv), ("gA_hc", GHC, gA_hc)):
        print(f"   bwd {name}: relL2 {rel(R.tp_decode(T, n), ref.float()):.3e}")
    print(f"   k_appear_bwd {timed(oldb):.1f} us   k_appear2_bwd {timed(newb):.1f} us")


run(1000)
run(100352)
