"""Dense SDF-volume evaluation for mesh extraction.

The reference's periodic evaluation sweeps a 512^3 / 768^3 grid through ``get_sdf_raw`` / ``get_shift_sdf_raw`` /
``get_sdf_vals`` in 100 000-point chunks with a host copy per chunk (utils/plots.py:154-205: 1 342 chunks at 512^3).
This helper keeps the grid on the device, uses chunks sized for 288 GB of HBM, and -- in ``mlp_precision='bf16'`` --
every chunk runs through the fused matrix-core kernel (csrc/sdf_mlp.hip).  Marching cubes itself is out of scope.
"""
import torch


def grid_axes(resolution, grid_boundary, device):
    lo, hi = grid_boundary
    return torch.linspace(lo, hi, resolution, device=device)


@torch.no_grad()
def evaluate_sdf_volume(implicit_network, resolution=512, grid_boundary=(-1.0, 1.0), kind="raw", chunk=1 << 22, device="cuda"):
    """Returns a [resolution^3, d_out] ('raw', 'shift') or [resolution^3, 1] ('min') float32 tensor on `device`;
    point order = meshgrid(x, y, z, indexing='ij') flattened, as the reference's get_grid_uniform."""
    fn = {"raw": implicit_network.get_sdf_raw, "shift": implicit_network.get_shift_sdf_raw, "min": implicit_network.get_sdf_vals}[kind]
    ax = grid_axes(resolution, grid_boundary, device)
    n = resolution ** 3
    out = None
    for start in range(0, n, chunk):
        idx = torch.arange(start, min(start + chunk, n), device=device)
        iz = idx % resolution
        iy = (idx // resolution) % resolution
        ix = idx // (resolution * resolution)
        vals = fn(torch.stack([ax[ix], ax[iy], ax[iz]], -1)).float()
        if out is None:
            out = torch.empty(n, vals.shape[1], device=device)
        out[start:start + vals.shape[0]] = vals
    return out
