"""Shared loaders for the golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def section(rec, prefix, as_torch=True):
    out = {}
    for k, v in rec.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(v) if as_torch else v
    return out


def oracle_cfg(rec, **extra):
    from oracle.stage1_oracle import Cfg
    m = {k[5:]: int(v) for k, v in rec.items() if k.startswith("meta.") and v.ndim == 0}
    beta = float(rec["state.density.beta"])
    S = m["S"]
    return Cfg(feature_vector_size=m["feat"], d_out=m["K"], dims=(m["width"],) * 2, render_dims=(m["width"],) * 2,
               num_levels=m["L"], base_size=m["base"], end_size=m["end"], logmap=m["logmap"], beta_init=beta,
               N_samples=S // 2, N_samples_eval=S, N_samples_extra=S // 4, **extra)


def rand_dict(rec):
    r = section(rec, "rand.")
    out = {k: v for k, v in r.items() if not k.startswith("bg.")}
    bg = {k[3:]: v for k, v in r.items() if k.startswith("bg.")}
    if bg:
        out["bg"] = bg
    if "bg_xy0" in out:
        out["bg_xy0"] = tuple(int(v) for v in out["bg_xy0"])
    return out


def stock_hash_table(seed, n_entries, C=2):
    """The embedding table of a hash_stock_s* fixture: its 12.2 M floats are regenerated from the seed instead of stored
    (torch's CPU generator is bit-reproducible; the fixture keeps a checksum and a strided sample to prove it)."""
    g = torch.Generator().manual_seed(1000 + seed)
    return (torch.rand(n_entries, C, generator=g) * 2 - 1) * 0.5


def load_hash_stock(name):
    """A hash_stock_s* fixture with its table regenerated and its sparse table gradients densified."""
    r = load(name)
    emb = stock_hash_table(int(r["seed"]), int(r["offsets"][-1]))
    assert abs(float(emb.double().sum()) - float(r["emb_checksum"])) < 1e-9 and np.array_equal(emb[::65537].numpy(), r["emb_sample"]), \
        "the regenerated table differs from the one the fixture was made with"
    out = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) and v.ndim > 0 else v) for k, v in r.items()}
    out["emb"] = emb
    for k in ("grad_emb", "grad2_emb"):
        dense = torch.zeros_like(emb)
        dense[out[k + "_rows"]] = out[k + "_vals"]
        out[k] = dense
    return out


# ------------------------------------------------------------------------------------------------ full-size fixtures (T = 2^19)
# A fixture at BASELINE configs[1]'s sizes cannot store 48.8 MB tables or their gradients.  Tables are regenerated from a seed (torch's
# CPU generator is bit-reproducible) and checked against a stored digest; table-sized RESULTS (gradients, parameters after Adam) are
# stored as the same digest: every TABLE_SAMPLE_STRIDE-th row (the stride is coprime to every level size, so each level is sampled
# at ~1 %) plus per-level float64 sums of g and g^2.
TABLE_KEYS = ("implicit_network.encoding.embeddings", "implicit_network.color_encoding.embeddings")
TABLE_SAMPLE_STRIDE = 97


def seeded_table(seed, n_entries, scale, C=2):
    g = torch.Generator().manual_seed(int(seed))
    return (torch.rand(n_entries, C, generator=g) * 2 - 1) * scale


def table_digest(t, offsets, stride=TABLE_SAMPLE_STRIDE):
    """{"vals", "level_sum", "level_sq"} of a [n, C] table-sized tensor (host); vals = rows 0, stride, 2 stride, ..."""
    t = t.detach().cpu()
    td = t.double()
    offsets = [int(o) for o in offsets]
    return {"vals": t[::stride].numpy().copy(),
            "level_sum": np.array([float(td[a:b].sum()) for a, b in zip(offsets[:-1], offsets[1:])]),
            "level_sq": np.array([float((td[a:b] ** 2).sum()) for a, b in zip(offsets[:-1], offsets[1:])])}


def load_full(name):
    """A full-size fixture with its two hash tables regenerated from their seeds (and verified against the stored digests)."""
    rec = load(name)
    from holoscene_amd.hashencoder.hashgrid import level_offsets
    m = {k[5:]: v for k, v in rec.items() if k.startswith("meta.")}
    scale = np.exp2(np.log2(int(m["end"]) / int(m["base"])) / (int(m["L"]) - 1))
    offs = level_offsets(3, int(m["L"]), scale, int(m["base"]), int(m["logmap"]))
    rec["aux.offsets"] = np.asarray(offs, dtype=np.int64)
    for i, k in enumerate(TABLE_KEYS):
        t = seeded_table(int(m["table_seed"]) + i, int(offs[-1]), float(m["emb_scale"]))
        dg = table_digest(t, offs, stride=TABLE_SAMPLE_STRIDE * 64)
        assert np.array_equal(dg["vals"], rec[f"state.{k}#vals"]) and np.allclose(dg["level_sum"], rec[f"state.{k}#level_sum"], rtol=0, atol=1e-9), \
            "the regenerated table differs from the one the fixture was made with"
        rec[f"state.{k}"] = t.numpy()
    return {k: v for k, v in rec.items() if not (k.startswith("state.") and "#" in k)}


def digest_sections(rec, prefix):
    """{table key: {"vals", "level_sum", "level_sq"}} of the digests stored under `prefix` ("grad.", "adam1.")."""
    out = {}
    for k, v in rec.items():
        if k.startswith(prefix) and "#" in k:
            name, field = k[len(prefix):].split("#")
            out.setdefault(name, {})[field] = v
    return out
