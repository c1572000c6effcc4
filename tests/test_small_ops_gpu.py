"""GPU: csrc/small_ops.hip (hs_assemble, hs_abs_shift) against their whole-tensor formulations -- exact where both sides add the same
fp32 numbers in the same order (term by term); long block reductions to summation-order tolerance."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _be():
    from holoscene_amd.hashencoder.backend import _backend
    return _backend


def test_assemble_terms_maps_and_reductions():
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    A = torch.randn(256, 128, generator=g).to(dev)
    B = torch.randn(256, 128, generator=g).to(dev)
    acc = torch.randn(544, generator=g).to(dev)
    part = torch.randn(512, 32, generator=g).to(dev)
    W2a, W2b = torch.randn(32, 256, generator=g).to(dev), torch.randn(32, 256, generator=g).to(dev)
    cols = torch.randperm(80, generator=g)[:71].to(torch.int32).to(dev)
    K = 21
    out = _be().assemble([
        ((256, 71), [(A, 128, cols)]),
        ((K, 256), [(W2a, 256, 0), (W2b, 256, 0)]),
        ((256, 1), [(acc, 1, 0), (B, 128, 80)]),
        ((256, 1), [(acc, 1, 256), (A, 128, 80)]),
        ((1, K), [(acc, 0, 512), (part, 0, 0, 512, 32)]),
        ((3, 5), [(A, 128, 7), (B, 128, 9), (acc, 0, 100)])])
    assert torch.equal(out[0], A.index_select(1, cols.long()))
    assert torch.equal(out[1], W2a[:K] + W2b[:K])
    assert torch.equal(out[2].view(-1), acc[:256] + B[:, 80])
    assert torch.equal(out[3].view(-1), acc[256:512] + A[:, 80])
    want = acc[512:512 + K].double() + part[:, :K].double().sum(0)       # (a wave per element: lanes add strided blocks, then meet)
    assert torch.allclose(out[4].view(-1).double(), want, rtol=0, atol=2e-5)
    assert torch.equal(out[5], A[:3, 7:12] + B[:3, 9:14] + acc[100:105])
    nine = _be().assemble([((2, 2), [(A, 128, i)]) for i in range(9)])        # more jobs than one launch holds
    assert all(torch.equal(t, A[:2, i:i + 2]) for i, t in enumerate(nine))
    with pytest.raises(RuntimeError):
        _be().assemble([((2, 2), [(A, 128, 0)] * 4)])


def test_density_beta_forward_backward():
    from holoscene_amd.model.density import LaplaceDensity
    for b0 in (0.1, -0.03, 0.0):
        d = LaplaceDensity({"beta": b0}, beta_min=1e-4).cuda()
        y = d.get_beta()
        assert torch.equal(y, d.beta.detach().abs() + d.beta_min)
        (y * 3.0).backward()
        assert torch.equal(d.beta.grad, 3.0 * torch.sgn(d.beta.detach()))
        with d.shared_beta() as s:
            assert d.get_beta() is s


def test_zero_pool_hands_out_untouched_zeros():
    from holoscene_amd.hashencoder import backend
    pool = torch.zeros(64, device="cuda")
    backend.set_zero_pool(pool)
    a, b = backend.zeros_small(10, "cuda"), backend.zeros_small(7, "cuda")
    assert a.data_ptr() == pool.data_ptr() and b.data_ptr() == pool[12:].data_ptr()       # 16-byte steps
    a.fill_(1.0)
    c = backend.zeros_small(100, "cuda")          # does not fit: a fresh tensor
    assert c.numel() == 100 and float(c.abs().sum()) == 0 and float(b.abs().sum()) == 0
    backend.set_zero_pool(None)
    assert backend.zeros_small(4, "cuda").data_ptr() != pool.data_ptr()


def test_iteration_prologue_and_epilogue_kernels():
    """csrc/iter_ops.hip: weight norms, beta, the uniform pool and the optimiser tick in one launch (hs_iter_prologue) against their
    whole-tensor formulations; the tail (hs_iter_epilogue) against autograd through torch._weight_norm and abs."""
    from holoscene_amd.hashencoder import backend as be
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    shapes = [(256, 71), (256, 256), (32, 256), (3, 337)]
    vs = [torch.randn(*s_, generator=g).to(dev) for s_ in shapes]
    gs = [torch.rand(s_[0], 1, generator=g).add(0.5).to(dev) for s_ in shapes]
    beta, beta_min = torch.tensor([-0.02], device=dev), torch.tensor([1e-4], device=dev)
    n = 208928 + 3
    pool = torch.full((n,), -1.0, device=dev)
    rs = torch.tensor([1234567, 0, 0], dtype=torch.int64, device=dev)
    st = be.hsAdamState()
    st.step = 4
    for i, v in enumerate((1e-2, 5e-4, 5e-4)):
        st.lr0[i] = v
    state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(dev)
    ref_state = state.clone()
    Ws, beta_eff = _be().iter_prologue(vs, gs, pool, rs, beta, beta_min, (state, 0.9, 0.99, 0.999))
    for W, v, g_ in zip(Ws, vs, gs):
        assert torch.allclose(W, torch._weight_norm(v, g_, 0), rtol=2e-6, atol=1e-7)
    assert torch.equal(beta_eff, beta.abs() + beta_min)
    _be().adam_tick(ref_state, 0.9, 0.99, 0.999)
    assert torch.equal(state, ref_state)
    assert rs.tolist() == [1234567, 1, 0]
    first = pool.clone()
    assert float(first.min()) >= 0.0 and float(first.max()) < 1.0
    assert abs(float(first.mean()) - 0.5) < 4 * (1 / 12 / n) ** 0.5 and abs(float(first.var()) - 1 / 12) < 1e-3
    assert first.unique().numel() > 0.98 * n      # 24-bit draws: ~0.6 % birthday collisions at this length
    _be().iter_prologue([], [], pool, rs)
    assert rs.tolist() == [1234567, 2, 0] and not torch.equal(pool, first)        # the next launch draws the next pool
    rs2 = torch.tensor([1234567, 0, 0], dtype=torch.int64, device=dev)
    pool2 = torch.empty(n, device=dev)
    _be().iter_prologue([], [], pool2, rs2)
    assert torch.equal(pool2, first)                                              # (seed, counter) name the pool
    # ---- tail
    gWs = [torch.randn_like(v) for v in vs]
    vr = [v.clone().requires_grad_() for v in vs]
    gr = [g_.clone().requires_grad_() for g_ in gs]
    br = beta.clone().requires_grad_()
    parts = [torch.randn(1024, device=dev), torch.randn(1, device=dev)]
    loss = sum((torch._weight_norm(v, g_, 0) * gW).sum() for v, g_, gW in zip(vr, gr, gWs)) + (br.abs() + beta_min).sum() * sum(p.sum() for p in parts)
    loss.backward()
    outs = [(torch.empty_like(v), torch.empty_like(g_)) for v, g_ in zip(vs, gs)]
    gb = torch.empty(1, device=dev)
    _be().iter_epilogue(vs, gs, gWs, outs, beta, parts, gb)
    for (gv, gg), v, g_ in zip(outs, vr, gr):
        assert torch.allclose(gv, v.grad, rtol=1e-4, atol=1e-5) and torch.allclose(gg, g_.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gb, br.grad, rtol=1e-5, atol=1e-5)


def test_iteration_prologue_context_equals_the_separate_contexts():
    """model/network.py: iteration_prologue() against density.shared_beta() + shared_effective_weights(): same beta, same matrices, same
    parameter gradients -- written into the flat gradient buffer's views without a copy --, one optimiser tick."""
    from holoscene_amd.model import network as net
    from holoscene_amd.training.trainer import Stage1Trainer, stock_conf
    tr = Stage1Trainer(stock_conf(beta=0.01, mlp_precision="bf16"), device="cuda", optimizer="flat", graph=False)
    m, flat = tr.model, tr.flat
    lins = m.weight_norm_layers()
    coef = [torch.randn_like(l.weight_v) for l in lins]

    def run(ctx):
        flat.zero_grad()
        with ctx as rng:
            Ws = net.effective_weights(lins)
            b = m.density.get_beta()
            loss = sum((W * c).sum() for W, c in zip(Ws, coef)) + 7.0 * b
            vals = [W.detach().clone() for W in Ws] + [b.detach().clone()]
        loss.backward()
        flat.gather_grads()
        return vals, flat.flat_g.clone(), rng

    import contextlib

    @contextlib.contextmanager
    def old():
        with m.density.shared_beta(), net.shared_effective_weights(lins):
            yield None

    v0, g0, _ = run(old())
    step0 = flat.read_state().step
    v1, g1, rng = run(net.iteration_prologue(m, flat, m.uniform_sizes(64)))
    assert all(torch.equal(a, b) for a, b in zip(v0, v1))
    assert torch.allclose(g0, g1, rtol=1e-6, atol=1e-8) and float(g1.abs().sum()) > 0
    assert flat.read_state().step == step0 + 1 and flat._ticked
    flat.end_update()
    assert set(rng) == set(m.uniform_sizes(64)) and all(0.0 <= float(t.min()) and float(t.max()) < 1.0 for t in rng.values())


def test_assemble_sums_bf16_partial_stacks_in_the_same_launch():
    """hs_assemble on the weight-gradient kernels' partial stacks (bf16 [slices, rows, ld]): slice sum + column selection / window /
    accumulation in one launch == hs_sum_slices followed by the fp32 assembly; quads, scalar units, misaligned windows, wide reductions."""
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    bf = torch.bfloat16
    st1 = torch.randn(98, 256, 256, generator=g).to(dev).to(bf)
    st0 = torch.randn(64, 256, 128, generator=g).to(dev).to(bf)
    st2 = torch.randn(55, 32, 256, generator=g).to(dev).to(bf)
    w2p = torch.randn(300, 32, 256, generator=g).to(dev)
    cs1 = torch.randn(98, 256, generator=g).to(dev)
    acc = torch.randn(544, generator=g).to(dev)
    cols = torch.randperm(80, generator=g)[:71].to(torch.int32).to(dev)
    K = 21
    win = torch.zeros(256, 337, device=dev)
    flat = torch.zeros(256, device=dev)
    out = _be().assemble([
        ((256, 256), [(st1, 256, 0, 98, 65536)]),
        ((256, 71), [(st0, 128, cols, 64, 256 * 128)]),
        ((K, 256), [(st2, 256, 0, 55, 32 * 256), (w2p, 256, 0, 300, 32 * 256)]),
        ((1, 256), [(acc, 0, 0), (cs1, 0, 0, 98, 256)], (flat, None)),
        ((256, 256), [(st1, 256, 0, 98, 65536)], (win, 81)),
        ((256, 81), [(st0, 128, 3, 64, 256 * 128)], (win, 0))])
    s1, s0, s2 = st1.float().sum(0), st0.float().sum(0), st2.float().sum(0)
    assert torch.allclose(out[0], s1, rtol=1e-5, atol=1e-4)
    assert torch.allclose(out[1], s0.index_select(1, cols.long()), rtol=1e-5, atol=1e-4)
    assert torch.allclose(out[2], s2[:K] + w2p.sum(0)[:K], rtol=1e-5, atol=2e-4)
    assert out[3] is flat and torch.allclose(flat, acc[:256] + cs1.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(win[:, 81:], s1, rtol=1e-5, atol=1e-4) and torch.allclose(win[:, :81], s0[:, 3:84], rtol=1e-5, atol=1e-4)
