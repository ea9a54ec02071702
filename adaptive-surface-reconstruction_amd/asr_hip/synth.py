"""Deterministic synthetic inputs and seeded weights.

There is no network access for Thingi10k scans or the released model.pt, so benchmarks and
parity tests use the generators below: C1/C2 = uniform unit sphere, C3 = scan-like union of
analytic shapes seen by ~50 virtual pinhole cameras (statistics of
models/v0/datareader.py:441-449,552-576), C5 = the same with 10x density variance.
"""
from collections import OrderedDict

import numpy as np


# ---------------------------------------------------------------------------------------------
# model parameter table: names / shapes of UNet5(with_importance='all', normalized_channels=8,
# residual_skip_connection=True).state_dict()  (models/v0/net_definitions_torch.py:390-499)
# ---------------------------------------------------------------------------------------------
def unet5_param_shapes(channel_div=1, normalized_channels=8):
    d = channel_div
    nc = normalized_channels
    c0 = 32 // d
    shapes = OrderedDict()
    shapes["cconv_block_in.conv1.kernel"] = (4, 4, 4, 4, c0)
    shapes["cconv_block_in.conv1.bias"] = (c0,)

    def conv(name, k, cin, cout):
        shapes[name + ".kernel"] = (k, cin, cout)
        shapes[name + ".bias"] = (cout,)

    def enc_block(name, cin, cout):
        assert nc < cout, "normalized_channels >= block width selects another reference branch"
        conv(name + ".conv1a", 55, cin, cout - nc)
        conv(name + ".conv1b", 55, cin, nc)
        for i in (2, 3, 4):
            conv(name + ".conv%d" % i, 55, cout, cout)

    def dec_block(name, cin, cout):
        conv(name + ".conv1", 55, cin, cout)
        for i in (2, 3, 4):
            conv(name + ".conv%d" % i, 55, cout, cout)

    def down(name, cin, cout):
        conv(name + ".conv1a", 9, cin, cout - nc)
        conv(name + ".conv1b", 9, cin, nc)

    e0, d1, e1, d2, e2, d3, e3, e4 = (64 // d, 128 // d, 128 // d, 256 // d, 256 // d, 256 // d,
                                      256 // d, 256 // d)
    enc_block("sparseconv_encblock0", c0, e0)
    down("sparseconv_down1", e0, d1)
    enc_block("sparseconv_encblock1", d1, e1)
    down("sparseconv_down2", e1, d2)
    enc_block("sparseconv_encblock2", d2, e2)
    down("sparseconv_down3", e2, d3)
    enc_block("sparseconv_encblock3", d3, e3)
    enc_block("sparseconv_encblock4", d3, e4)  # fed by the re-used down3 (:456-460)
    u3 = 256 // d
    conv("sparseconv_up3.conv1", 9, e4, u3)
    dec_block("sparseconv_decblock3", u3 + e3, 256 // d)
    u2 = 256 // d
    conv("sparseconv_up2.conv1", 9, 256 // d, u2)
    dec_block("sparseconv_decblock2", u2 + e2, 256 // d)
    u1 = 256 // d
    conv("sparseconv_up1.conv1", 9, 256 // d, u1)
    dec_block("sparseconv_decblock1", u1 + e1, 128 // d)
    u0 = 64 // d
    conv("sparseconv_up0.conv1", 9, 128 // d, u0)
    dec_block("sparseconv_decblock0", u0, 32 // d)
    # the reference hard-codes in_features = 32 + 3 (:486); only channel_div == 1 can decode there
    code = 32 // d
    shapes["dense_decoder1.weight"] = (32 // d, 3 + code)
    shapes["dense_decoder1.bias"] = (32 // d,)
    shapes["dense_decoder2.weight"] = (32 // d, 32 // d)
    shapes["dense_decoder2.bias"] = (32 // d,)
    shapes["dense_decoder3.weight"] = (2, 32 // d)
    return shapes


def make_weights(channel_div=1, seed=0, init="variance"):
    """Seeded weights under the reference tensor names.

    init="variance": variance preserving (the reference initialisers collapse the
    forward to std ~3e-7, which would make any absolute tolerance vacuous).
    init="reference": uniform(-0.05, 0.05) kernels, zero bias (models/common_torch.py:57-58);
    used for timing only (timing does not depend on the values).
    Values come from numpy's PCG64 so fixtures and tests regenerate them bit-identically.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in unet5_param_shapes(channel_div).items():
        if init == "reference":
            if name.endswith(".kernel") or name.endswith(".weight"):
                w = rng.uniform(-0.05, 0.05, size=shape)
            else:
                w = np.zeros(shape)
        else:
            if name.endswith(".bias"):
                w = rng.standard_normal(shape) * 0.05
            elif name.startswith("cconv"):
                w = rng.standard_normal(shape) * np.sqrt(2.0 / shape[3])
            elif name.endswith(".kernel"):
                k, cin, _ = shape
                # ~8 of 55 slots are occupied per voxel; conv1b is an importance weighted mean
                fan = cin if name.endswith("conv1b.kernel") else cin * (8.0 if k == 55 else 1.0)
                w = rng.standard_normal(shape) * np.sqrt(2.0 / fan)
            else:  # torch Linear [out, in]
                w = rng.standard_normal(shape) * np.sqrt(2.0 / shape[1])
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# ---------------------------------------------------------------------------------------------
# point clouds
# ---------------------------------------------------------------------------------------------
def sphere_cloud(n, seed=0):
    """C1 / C2: n points on the unit sphere, normals = positions."""
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    p = p.astype(np.float32)
    return p, p.copy()


def fused_scan_cloud(num_scans, points_per_scan, seed=0, device="cpu", spacing=4.0):
    """C4: `num_scans` disjoint C3-style scans fused into one cloud -- scan i (seed + i) is moved to
    cell i of a 2 x 2 x 2 .. lattice with `spacing` between the scene centres (a scene spans less than +-2).
    Returns (points, normals) on `device`."""
    import torch
    pts, nrm = [], []
    side = 1
    while side ** 3 < num_scans:
        side += 1
    for i in range(num_scans):
        p, q = scan_cloud(points_per_scan, seed=seed + i, device=device)
        off = torch.tensor([i % side, (i // side) % side, i // (side * side)], dtype=torch.float32, device=p.device)
        pts.append(p + off * spacing)
        nrm.append(q)
    return torch.cat(pts).contiguous(), torch.cat(nrm).contiguous()


def knn_radii(points, k=24):
    """radius_i = distance to the k-th nearest neighbour including the point itself
    (cpp/lib/nsearch.cpp:30-51); exact, scipy cKDTree on the host cores."""
    from scipy.spatial import cKDTree
    pts = np.asarray(points)
    t = cKDTree(pts)
    d, _ = t.query(pts, k=k, workers=-1)
    return d[:, -1].astype(np.float32)


def knn_radii_gpu(points, k=24):
    """the same radii on the GPU (asr_hip_knn_radius, exact): points is a CUDA tensor"""
    from . import _lib, ops
    mn, mx = bounding_box(points, 1e-3)
    return ops.knn_radius(_lib.frame_init(mn, mx), points, k)


def bounding_box(points, margin=0.1):
    """exact bbox widened by `margin` (models/v0/datareader.py:225-227)"""
    import torch
    if isinstance(points, torch.Tensor):
        # reduce over the contiguous dimension of a transposed copy: torch's reduction along dim 0 of an [N, 3]
        # tensor takes 7 ms per call at 10 M points, this 0.3 ms
        t = points.t().contiguous()
        both = torch.stack([t.amin(dim=1), t.amax(dim=1)]).cpu().numpy()
        mn, mx = both[0], both[1]
    else:
        mn, mx = points.min(0), points.max(0)
    m = np.float32(margin)
    return (mn.astype(np.float32) - m), (mx.astype(np.float32) + m)


def _scene_sdf(p):
    """union of sphere, torus, rounded box and a thin slab; p [M,3] torch tensor"""
    import torch
    q = p - p.new_tensor([-0.9, 0.0, 0.0])
    sphere = q.norm(dim=1) - 0.7
    q = p - p.new_tensor([0.9, 0.0, 0.0])
    t = torch.stack([torch.sqrt(q[:, 0] ** 2 + q[:, 2] ** 2) - 0.6, q[:, 1]], 1)
    torus = t.norm(dim=1) - 0.25
    q = (p - p.new_tensor([0.0, 0.1, 1.3])).abs() - p.new_tensor([0.5, 0.4, 0.3])
    box = q.clamp(min=0).norm(dim=1) + q.max(dim=1).values.clamp(max=0) - 0.1
    q = (p - p.new_tensor([0.0, -0.95, 0.3])).abs() - p.new_tensor([1.8, 0.02, 1.4])
    slab = q.clamp(min=0).norm(dim=1) + q.max(dim=1).values.clamp(max=0)
    return torch.minimum(torch.minimum(sphere, torus), torch.minimum(box, slab))


def scan_cloud(n, seed=0, device="cpu", density_variance=1.0, num_cameras=50):
    """C3 / C5: scan-like cloud. Returns (points, normals) float32 torch tensors on `device`.

    Rays from `num_cameras` pinhole cameras on a sphere around the scene are sphere-traced
    against the analytic scene; depth gets Laplace noise 0.001*d, 5 % of the pixels are dropped.
    density_variance > 1 spreads the camera distances so that the sample density varies by about
    that factor (C5).
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dev = torch.device(device)
    pts, nrm = [], []
    have = 0
    cam_dirs = torch.randn(num_cameras, 3, generator=g, device=dev)
    cam_dirs = cam_dirs / cam_dirs.norm(dim=1, keepdim=True)
    cam_dirs[:, 1] = cam_dirs[:, 1].abs() * 0.8 + 0.1  # above the slab
    cam_dirs = cam_dirs / cam_dirs.norm(dim=1, keepdim=True)
    spread = float(np.sqrt(density_variance))
    cam_dist = 4.0 * torch.exp((torch.rand(num_cameras, generator=g, device=dev) - 0.5) *
                               2.0 * np.log(spread)) if spread > 1 else torch.full(
                                   (num_cameras,), 4.0, device=dev)
    rounds = 0
    rate = 0.5  # fraction of rays that become points; refined after the first round
    while have < n and rounds < 12:
        rounds += 1
        per_cam = int((n - have) * 1.15 / rate / num_cameras) + 64
        m = per_cam * num_cameras
        cam = torch.arange(num_cameras, device=dev).repeat_interleave(per_cam)
        origin = cam_dirs[cam] * cam_dist[cam, None]
        fwd = -cam_dirs[cam]
        up = torch.tensor([0.0, 1.0, 0.0], device=dev).expand(m, 3)
        right = torch.cross(fwd, up, dim=1)
        right = right / right.norm(dim=1, keepdim=True).clamp(min=1e-6)
        upv = torch.cross(right, fwd, dim=1)
        uv = (torch.rand(m, 2, generator=g, device=dev) - 0.5) * 1.2  # ~62 degree fov
        dirs = fwd + uv[:, :1] * right + uv[:, 1:] * upv
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        t = torch.zeros(m, device=dev)
        for _ in range(96):
            d = _scene_sdf(origin + dirs * t[:, None])
            t = t + d
        p = origin + dirs * t[:, None]
        hit = (_scene_sdf(p).abs() < 1e-3) & (t < 20.0)
        keep = hit & (torch.rand(m, generator=g, device=dev) > 0.05)
        p, t, dirs_k = p[keep], t[keep], dirs[keep]
        # normals from the sdf gradient at the noise-free hit
        eps = 1e-3
        e = torch.eye(3, device=dev) * eps
        grad = torch.stack([_scene_sdf(p + e[i]) - _scene_sdf(p - e[i]) for i in range(3)], 1)
        nn_ = grad / grad.norm(dim=1, keepdim=True).clamp(min=1e-9)
        u = (torch.rand(p.shape[0], generator=g, device=dev) - 0.5).clamp(-0.4999999, 0.4999999)  # log1p(-1) = -inf
        lap = -torch.sign(u) * torch.log1p(-2 * u.abs())
        p = p + dirs_k * (0.001 * t * lap)[:, None]
        pts.append(p)
        nrm.append(nn_)
        have += p.shape[0]
        rate = max(0.05, p.shape[0] / float(m))
    points = torch.cat(pts)[:n].contiguous().float()
    normals = torch.cat(nrm)[:n].contiguous().float()
    if points.shape[0] < n:
        raise RuntimeError("scan_cloud: scene produced too few hits")
    perm = torch.randperm(n, generator=g, device=dev)
    return points[perm].contiguous(), normals[perm].contiguous()
