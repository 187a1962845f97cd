"""Synthetic 480x640 RGB-D pairs for tests and bench.py (no dataset / renderer offline).

Follows the recipe of SURVEY.md §8d: seed 2333 (toolkit/LM6d_1_gen_rendered_pose.py:20), LINEMOD
intrinsics (deepim/config/config.py:58), target pose = random rotation + t in the LINEMOD range,
source pose = target perturbed by Euler noise N(0,15°) (reject > 45°) and xyz noise N(0, 1/1/5 cm)
(toolkit/LM6d_1_gen_rendered_pose.py:54,86-112); the "object" is a textured ellipsoid (≈0.1 m, the
size of LINEMOD 'ape') ray-cast analytically into image / depth / mask; tensors are laid out the way
lib/pair_matching/data_pair.py + lib/utils/image.py:583-594 hand them to the network (RGB order,
mean-subtracted, NCHW float32; mask_observed = filled box of the rendered mask, INIT_MASK box_rendered).
Host-side numpy only — this is input staging, not part of the timed path.
"""
import numpy as np

K_LINEMOD = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], dtype=np.float32)
PIXEL_MEANS = np.array([123.68, 116.779, 103.939], dtype=np.float32)


def euler_to_mat(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def sample_pose_pair(rng, K=K_LINEMOD, H=480, W=640):
    """(pose_tgt, pose_src) 3x4 float32; src = perturbed tgt, centre ≥ 16 px inside the frame."""
    while True:
        R = random_rotation(rng)
        t = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1), rng.uniform(0.6, 1.2)])
        ang = rng.normal(0, 15.0, 3)
        if np.any(np.abs(ang) > 45):
            continue
        Rs = euler_to_mat(*np.deg2rad(ang)) @ R
        ts = t + rng.normal(0, 1, 3) * np.array([0.01, 0.01, 0.05])
        c = K.astype(np.float64) @ ts
        if not (16 < c[0] / c[2] < W - 16 and 16 < c[1] / c[2] < H - 16):
            continue
        tgt = np.concatenate([R, t[:, None]], 1).astype(np.float32)
        src = np.concatenate([Rs, ts[:, None]], 1).astype(np.float32)
        return tgt, src


def raycast_ellipsoid(pose, axes, K=K_LINEMOD, H=480, W=640):
    """-> (rgb uint8-valued float32 (H,W,3), depth float32 (H,W) in metres, 0 = background)."""
    R, t = pose[:, :3].astype(np.float64), pose[:, 3].astype(np.float64)
    Kinv = np.linalg.inv(K.astype(np.float64))
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = np.stack([u, v, np.ones_like(u)], -1) @ Kinv.T          # rays, d_z = 1
    dm = d @ R                                                    # R^T d
    om = -(R.T @ t)
    S = 1.0 / np.asarray(axes, np.float64)
    a = np.sum((dm * S) ** 2, -1)
    b = 2 * np.sum((dm * S) * (om * S), -1)
    c = np.sum((om * S) ** 2) - 1
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0))) / (2 * a), 0.0)
    hit &= s > 0
    depth = np.where(hit, s, 0.0).astype(np.float32)
    pm = om + dm * s[..., None]                                   # model-frame surface point
    nrm = pm * S * S
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-12)
    lam = np.clip(0.35 + 0.65 * np.abs(np.sum((nrm @ R.T) * np.array([0.3, -0.4, -0.86]), -1)), 0, 1)
    ph = pm / np.asarray(axes)
    tex = np.stack([0.5 + 0.5 * np.sin(9 * ph[..., 0] + 2 * ph[..., 1]), 0.5 + 0.5 * np.sin(7 * ph[..., 1] - 3 * ph[..., 2]),
                    0.5 + 0.5 * np.sin(11 * ph[..., 2] + 5 * ph[..., 0])], -1)
    rgb = np.where(hit[..., None], np.floor(255 * tex * lam[..., None]), 0).astype(np.float32)
    return rgb, depth


def ellipsoid_mesh(axes, n_lat=48, n_lon=96):
    """Triangulated ellipsoid for the rasteriser tests / bench (LINEMOD 'ape' has ~5.8k vertices, ~11.7k faces;
    the default gives 4.7k / 9.0k). Returns dict(vertices (V,3), faces (F,3) int32, uv (V,2), colors (V,3) 0..255)."""
    th = np.linspace(0.0, np.pi, n_lat + 1)                 # polar
    ph = np.linspace(0.0, 2 * np.pi, n_lon + 1)             # azimuth, seam duplicated so uv stays continuous
    T, P = np.meshgrid(th, ph, indexing="ij")
    unit = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3)
    vertices = (unit * np.asarray(axes, np.float64)).astype(np.float32)
    uv = np.stack([P / (2 * np.pi), T / np.pi], -1).reshape(-1, 2).astype(np.float32)
    colors = np.floor(255 * (0.5 + 0.5 * np.sin(np.stack([9 * unit[:, 0] + 2 * unit[:, 1], 7 * unit[:, 1] - 3 * unit[:, 2],
                                                           11 * unit[:, 2] + 5 * unit[:, 0]], -1)))).astype(np.float32)
    faces = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * (n_lon + 1) + j, i * (n_lon + 1) + j + 1
            c, d = a + n_lon + 1, b + n_lon + 1
            if i > 0:
                faces.append([a, c, b])
            if i < n_lat - 1:
                faces.append([b, c, d])
    return {"vertices": vertices, "faces": np.asarray(faces, np.int32), "uv": uv, "colors": colors}


def procedural_texture(h=256, w=512, seed=7):
    """(h,w,3) float32 texture on the 0..255 scale (integer-valued like a PNG), row 0 = v 0."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    base = np.stack([0.5 + 0.5 * np.sin(18 * x + 5 * y), 0.5 + 0.5 * np.sin(13 * y - 4 * x), 0.5 + 0.5 * np.sin(23 * x * y + 1)], -1)
    return np.floor(np.clip(255 * base + rng.uniform(-12, 12, (h, w, 3)), 0, 255)).astype(np.float32)


def to_tensor(rgb, means=PIXEL_MEANS):
    """HxWx3 RGB floats -> (3,H,W) tensor the way lib/utils/image.py:583-594 builds it from a BGR image:
    tensor[i] = bgr[..., 2-i] - pixel_means[2-i], i.e. RGB channel order with the config means REVERSED
    (which is why the Zoom* Props reverse `pixel_means`, zoom_image_with_factor.py:79-81)."""
    return np.ascontiguousarray((rgb - means[::-1].reshape(1, 1, 3)).transpose(2, 0, 1), dtype=np.float32)


def box_mask(mask):
    """INIT_MASK / UPDATE_MASK "box_rendered" (lib/utils/image.py:363-372): [y_start:y_end, x_start:x_end] with end = the
    LAST non-zero row/column used as an exclusive slice bound, so that row and column stay 0."""
    ys, xs = np.nonzero(mask)
    out = np.zeros_like(mask, dtype=np.float32)
    if len(ys):
        out[ys.min():ys.max(), xs.min():xs.max()] = 1
    return out


def sample_model_points(rng, axes, n=3000):
    p = rng.standard_normal((n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return np.ascontiguousarray((p * np.asarray(axes)).T, dtype=np.float32)  # (3, n)


def make_batch(B, seed=2333, H=480, W=640, n_frames=1, with_depth=True, n_points=3000, occlude=False):
    """Dict of numpy arrays for B pairs. `image_rendered/mask_rendered/depth_rendered/src_pose` carry a
    leading frame axis (n_frames, B, ...): frame 0 = the perturbed initial pose, frame i = a pose i/n of
    the way toward the target (pre-staged stand-ins for the re-rendered frames of iterations 2..n)."""
    rng = np.random.default_rng(seed)
    K = K_LINEMOD
    out = {k: [] for k in ("image_observed", "mask_observed", "depth_gt_observed", "pose_tgt", "point_cloud_model")}
    frames = {k: [[] for _ in range(n_frames)]
              for k in ("image_rendered", "mask_rendered", "depth_rendered", "src_pose", "mask_observed_frames")}
    for b in range(B):
        axes = np.array([0.05, 0.04, 0.035]) * rng.uniform(0.85, 1.15, 3)
        tgt, src = sample_pose_pair(rng, K, H, W)
        rgb_o, dep_o = raycast_ellipsoid(tgt, axes, K, H, W)
        bg = np.floor(rng.uniform(0, 255, (H, W, 3))).astype(np.float32)
        obs = np.where((dep_o > 0)[..., None], rgb_o, bg)
        if occlude:
            y0, x0 = rng.integers(0, H - 60), rng.integers(0, W - 60)
            obs[y0:y0 + 60, x0:x0 + 60] = np.floor(rng.uniform(0, 255, 3))
        out["image_observed"].append(to_tensor(obs))
        out["depth_gt_observed"].append(dep_o[None])
        out["pose_tgt"].append(tgt)
        out["point_cloud_model"].append(sample_model_points(rng, axes, n_points))
        for f in range(n_frames):
            a = f / float(n_frames)
            pose_f = src.copy()
            if f > 0:  # crude interpolation toward the target (re-orthonormalised)
                M = (1 - a) * src[:, :3].astype(np.float64) + a * tgt[:, :3].astype(np.float64)
                U, _, Vt = np.linalg.svd(M)
                pose_f[:, :3] = (U @ Vt).astype(np.float32)
                pose_f[:, 3] = (1 - a) * src[:, 3] + a * tgt[:, 3]
            rgb_r, dep_r = raycast_ellipsoid(pose_f, axes, K, H, W)
            frames["image_rendered"][f].append(to_tensor(rgb_r))
            frames["depth_rendered"][f].append(dep_r[None])
            frames["mask_rendered"][f].append((dep_r > 0.2).astype(np.float32)[None])
            frames["src_pose"][f].append(pose_f)
            # UPDATE_MASK=box_rendered: the observed mask is re-derived from each re-rendered frame
            frames["mask_observed_frames"][f].append(box_mask(frames["mask_rendered"][f][-1][0])[None])
        out["mask_observed"].append(box_mask(frames["mask_rendered"][0][-1][0])[None])
    res = {k: np.ascontiguousarray(np.stack(v), dtype=np.float32) for k, v in out.items()}
    for k, v in frames.items():
        res[k] = np.ascontiguousarray(np.stack([np.stack(x) for x in v]), dtype=np.float32)
    res["K"] = K.copy()
    if not with_depth:
        res.pop("depth_gt_observed")
        res.pop("depth_rendered")
    return res
