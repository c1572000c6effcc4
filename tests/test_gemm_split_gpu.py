"""GPU: fp32 products as sums of bf16 plane products (csrc/gemm_split.hip) against float64: three planes reproduce an fp32 GEMM's error,
two planes 16 mantissa bits; ragged shapes, strided rows, bias, the split-M weight-gradient form."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _err(c, want):
    return float((c.double() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (4099, 32, 71), (130, 257, 337), (257, 3, 256), (64, 128, 32), (5, 1, 7)])
def test_nt_matches_float64(M, N, K):
    from holoscene_amd.hashencoder.backend import _backend as be
    g = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) * 0.3).to(DEV), torch.randn(N, generator=g).to(DEV)
    want = a.double() @ b.double().t() + bias.double()
    lib = _err(torch.addmm(bias, a, b.t()), want)
    e3, e2 = _err(be.gemm_split_nt(a, b, bias, planes=3), want), _err(be.gemm_split_nt(a, b, bias, planes=2), want)
    print(f"PARITY gemm_split nt {M}x{N}x{K}: library fp32 {lib:.2e}, 3 planes {e3:.2e}, 2 planes {e2:.2e}")
    assert e3 < max(2 * lib, 1e-6), "three planes: the error of an fp32 GEMM"
    assert e2 < 3e-5
    # strided rows (a column window of a wider matrix), no bias
    wide = torch.randn(M, K + 9, generator=g).to(DEV)
    assert _err(be.gemm_split_nt(wide[:, 5:5 + K], b), wide[:, 5:5 + K].double() @ b.double().t()) < max(2 * lib, 1e-6)


@pytest.mark.parametrize("M,N,K,S", [(4096, 256, 256, 8), (4099, 32, 71, 5), (1000, 257, 80, 3), (77, 3, 256, 4), (33, 256, 337, 1)])
def test_tn_partials_sum_to_the_weight_gradient(M, N, K, S):
    from holoscene_amd.hashencoder.backend import _backend as be
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, N, generator=g).to(DEV), torch.randn(M, K, generator=g).to(DEV)
    want = a.double().t() @ b.double()
    lib = _err(a.t() @ b, want)
    p3, p2 = be.gemm_split_tn(a, b, S, planes=3), be.gemm_split_tn(a, b, S, planes=2)
    assert p3.shape == (S, N, K)
    e3, e2 = _err(p3.sum(0), want), _err(p2.sum(0), want)
    print(f"PARITY gemm_split tn {M}x{N}x{K}/{S}: library fp32 {lib:.2e}, 3 planes {e3:.2e}, 2 planes {e2:.2e}")
    assert e3 < max(2 * lib, 1e-6) and e2 < 3e-5
