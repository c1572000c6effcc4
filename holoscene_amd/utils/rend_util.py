"""Camera helpers on the Stage-1 path (reference: utils/rend_util.py:9-17, 56-98, 112-125).

Unlike the reference, ``get_camera_params`` never writes into ``uv``: the reference adds
``ray_offset`` through a view of ``uv`` (rend_util.py:70-75), which makes a second call see twice
the offset.  ``HoloSceneNetwork.forward`` reproduces that effect explicitly.
"""
import torch
import torch.nn.functional as F


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:  # [-1,1] -> [0,1]
        img1 = (img1 + 1.0) / 2.0
        img2 = (img2 + 1.0) / 2.0
    mse = torch.mean((img1 - img2) ** 2)
    return -10.0 * torch.log10(mse)


def quat_to_rot(q):
    q = F.normalize(q, dim=1)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.view(-1, 3, 3)


def lift(x, y, z, intrinsics):
    """Pixel (x, y) at depth z -> homogeneous camera-space point, honouring the skew term."""
    intrinsics = intrinsics.to(x.device)
    fx = intrinsics[:, 0, 0, None]
    fy = intrinsics[:, 1, 1, None]
    cx = intrinsics[:, 0, 2, None]
    cy = intrinsics[:, 1, 2, None]
    sk = intrinsics[:, 0, 1, None]
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics, ray_offset=None):
    """uv [1,R,2], pose [1,4,4] (or [1,7] quaternion+translation), intrinsics [1,4,4]
    -> unit ray directions [1,R,3] in world space and camera centre [1,3]."""
    if pose.shape[1] == 7:
        cam_loc = pose[:, 4:]
        p = torch.eye(4, device=pose.device, dtype=pose.dtype).repeat(pose.shape[0], 1, 1)
        p[:, :3, :3] = quat_to_rot(pose[:, :4])
        p[:, :3, 3] = cam_loc
    else:
        cam_loc = pose[:, :3, 3]
        p = pose
    x_cam, y_cam = uv[:, :, 0], uv[:, :, 1]
    if ray_offset is not None:
        x_cam = x_cam + ray_offset[:, :, 0]
        y_cam = y_cam + ray_offset[:, :, 1]
    pts = lift(x_cam, y_cam, torch.ones_like(x_cam), intrinsics).permute(0, 2, 1)
    world = torch.bmm(p, pts).permute(0, 2, 1)
    world = world[..., :3] / world[..., 3:4]
    return F.normalize(world - cam_loc[:, None, :], dim=2), cam_loc


def get_sphere_intersections(cam_loc, ray_directions, r=1.0):
    """Depths [n, 2] (entry, exit; clamped at 0) at which rays from cam_loc [n, 3] along unit ray_directions [n, 3] cross the sphere of
    radius r about the origin (utils/rend_util.py:169-185 of the reference).  The reference prints 'BOUNDING SPHERE PROBLEM!' and
    exits the process when a ray misses the sphere; here that is a RuntimeError with the same words."""
    ray_cam_dot = (ray_directions.reshape(-1, 3) * cam_loc.reshape(-1, 3)).sum(-1, keepdim=True)
    under_sqrt = ray_cam_dot ** 2 - (cam_loc.reshape(-1, 3).norm(2, 1, keepdim=True) ** 2 - r ** 2)
    if bool((under_sqrt <= 0).any()):
        raise RuntimeError("BOUNDING SPHERE PROBLEM!")
    signs = torch.tensor([-1.0, 1.0], device=cam_loc.device)
    return (torch.sqrt(under_sqrt) * signs - ray_cam_dot).clamp_min(0.0)
