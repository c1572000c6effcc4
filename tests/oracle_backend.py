"""TEST-ONLY stand-in for holoscene_amd.hashencoder.backend._backend that runs the hash encoder on the
CPU oracle, so the host-side model logic can be exercised by `-m "not gpu"` tests.  The product never
imports this; on a GPU box the real backend (libholoscene_hip.so) is what runs."""
import torch

from oracle import hash_oracle


def _lbc(t, B, L, C):  # point-major [B, L*C] -> level-major [L,B,C]
    return t.view(B, L, C).permute(1, 0, 2).contiguous()


class OracleBackend:
    @staticmethod
    def fwd(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx):
        out, j = hash_oracle.fwd(inputs.contiguous(), embeddings.detach().contiguous(), offsets, S, H, dy_dx is not None)
        outputs.copy_(out.permute(1, 0, 2).reshape(B, L * C))
        if dy_dx is not None:
            dy_dx.copy_(j.view(B, L, D * C).permute(1, 0, 2))

    @staticmethod
    def bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs):
        j = dy_dx.permute(1, 0, 2).reshape(B, -1).contiguous() if dy_dx is not None else torch.empty(1)
        emb_shape = (int(offsets[-1]), C)
        gx, ge = hash_oracle.bwd(_lbc(grad, B, L, C), inputs, torch.zeros(emb_shape), offsets, S, H, grad_inputs is not None, j)
        if grad_embeddings is not None:
            grad_embeddings.add_(ge)
        if grad_inputs is not None:
            grad_inputs.copy_(gx)

    @staticmethod
    def bwd2(grad, inputs, offsets, B, D, C, L, S, H, dy_dx, ggx, grad_grad, grad2_embeddings):
        j = dy_dx.permute(1, 0, 2).reshape(B, -1).contiguous()
        gg, g2 = hash_oracle.bwd2(_lbc(grad, B, L, C), inputs, torch.zeros(int(offsets[-1]), C), offsets, S, H, j, ggx)
        if grad_grad is not None:
            grad_grad.copy_(gg.permute(1, 0, 2).reshape(B, L * C))
        if grad2_embeddings is not None:
            grad2_embeddings.add_(g2)

    @staticmethod
    def bwd_jac(g_feat, g_dydx, inputs, offsets, grad_embeddings, B, D, C, L, S, H):
        emb0 = torch.zeros(int(offsets[-1]), C)
        if g_feat is not None:
            _, ge = hash_oracle.bwd(_lbc(g_feat, B, L, C), inputs, emb0, offsets, S, H, False, torch.empty(1))
            grad_embeddings.add_(ge)
        if g_dydx is not None:  # general cotangent = sum over d of rank-one terms G[:,:,d,:] (x) e_d
            G = g_dydx.view(L, B, D, C)
            dummy = torch.zeros(B, L * D * C)
            for d in range(D):
                e = torch.zeros(B, D)
                e[:, d] = 1
                _, g2 = hash_oracle.bwd2(G[:, :, d, :].contiguous(), inputs, emb0, offsets, S, H, dummy, e)
                grad_embeddings.add_(g2)

    # elementwise trunk stages: plain torch restatement (test-only)
    @staticmethod
    def softplus_tangent_fwd(A, bias, out):
        v = A[:, 0] + bias
        out[:, 0] = torch.nn.functional.softplus(v, beta=100)
        out[:, 1:] = torch.sigmoid(100 * v).unsqueeze(1) * A[:, 1:]

    @staticmethod
    def softplus_tangent_bwd(A, bias, G, gA, gbias):
        v = A[:, 0] + bias
        s = torch.sigmoid(100 * v)
        gA[:, 1:] = s.unsqueeze(1) * G[:, 1:]
        gA[:, 0] = s * G[:, 0] + 100 * s * (1 - s) * (A[:, 1:] * G[:, 1:]).sum(1)
        if gbias is not None:
            gbias.add_(gA[:, 0].sum(0))

    # fused input builders: plain torch restatement (test-only)
    @staticmethod
    def _pe(v, nfreq):
        parts, jac = [v], [torch.diag_embed(torch.ones_like(v))]
        for k in range(nfreq):
            f = 2.0 ** k
            s, c = torch.sin(v * f), torch.cos(v * f)
            parts += [s, c]
            jac += [torch.diag_embed(c * f), torch.diag_embed(-s * f)]
        return torch.cat(parts, -1), torch.cat(jac, -1)

    @classmethod
    def trunk_input_fwd(cls, x, feat, dydx, out, nfreq, L, C, jac_scale):
        B = x.shape[0]
        pe, pj = cls._pe(x, nfreq)
        fj = dydx.view(L, B, 3, C).permute(1, 2, 0, 3).reshape(B, 3, L * C) * jac_scale
        out[:, 0] = torch.cat([pe, feat], -1).to(out.dtype)
        out[:, 1:] = torch.cat([pj, fj], -1).to(out.dtype)

    @staticmethod
    def trunk_input_bwd(G, g_feat, g_dydx, nfreq, L, C, jac_scale):
        B = G.shape[0]
        P = 3 + 6 * nfreq
        g_feat.copy_(G[:, 0, P:].float())
        g_dydx.copy_((G[:, 1:, P:].float() * jac_scale).reshape(B, 3, L, C).permute(2, 0, 1, 3).reshape(L, B, 3 * C))

    @classmethod
    def render_input_fwd(cls, points, dirs, normals, fv, out, nfreq):
        out.copy_(torch.cat([cls._pe(points, nfreq)[0], cls._pe(dirs, nfreq)[0], cls._pe(normals, nfreq)[0], fv.float()], -1).to(out.dtype))

    @classmethod
    def render_input_bwd(cls, G, normals, d_normals, d_fv, nfreq, Fv):
        P = 3 + 6 * nfreq
        _, pj = cls._pe(normals, nfreq)                      # [B,3,P]
        d_normals.copy_(torch.einsum("bdp,bp->bd", pj, G[:, 2 * P:3 * P].float()))
        if d_fv is not None:
            d_fv.copy_(G[:, 3 * P:])


def install(monkeypatch):
    from holoscene_amd.hashencoder import backend
    monkeypatch.setattr(backend, "_backend", OracleBackend)
