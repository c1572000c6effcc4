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
