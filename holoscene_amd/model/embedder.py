"""NeRF positional encoding (reference: model/embedder.py:5-50).

Output order is the reference's: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...].
``embed_jacobian`` additionally returns d(embedding)/dx, which the value+Jacobian SDF trunk uses
instead of autograd's reverse passes.
"""
import torch


class Embedder:
    def __init__(self, multires, input_dims=3):
        self.multires = multires
        self.input_dims = input_dims
        self.out_dim = input_dims * (1 + 2 * multires)
        self.freqs = [2.0 ** k for k in range(multires)]  # log-sampled, max_freq_log2 = multires-1

    def embed(self, x):
        parts = [x]
        for f in self.freqs:
            parts.append(torch.sin(x * f))
            parts.append(torch.cos(x * f))
        return torch.cat(parts, -1)

    def embed_jacobian(self, x):
        """x [B,d] -> (emb [B,out], jac [B,d,out]) with jac[b,i,:] = d emb / d x_i."""
        d = x.shape[-1]
        eye = torch.eye(d, device=x.device, dtype=x.dtype).expand(x.shape[0], d, d)
        parts, jparts = [x], [eye]
        for f in self.freqs:
            s, c = torch.sin(x * f), torch.cos(x * f)
            parts += [s, c]
            jparts += [torch.diag_embed(c * f), torch.diag_embed(s * (-f))]
        return torch.cat(parts, -1), torch.cat(jparts, -1)


def get_embedder(multires, input_dims=3):
    obj = Embedder(multires, input_dims)
    return obj.embed, obj.out_dim
