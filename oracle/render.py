"""CPU restatement of the re-render step (TEST INFRASTRUCTURE — only tests/, smoke() and bench's cpu_baseline
may import this).

Follows lib/render_glumpy/render_py_multi.py:101-147 of the reference: pinhole projection with u0 = cx + 0.5
against GL pixel centres at +0.5 (so pixel index (i, j) is covered when it lies inside the triangle projected
with plain K), GL_LESS depth test with no culling (:93-95), metric depth with 0 background (:126-128), texture
colour * 255 (:123-125).  The reference draws through OpenGL, which cannot run here and whose sub-pixel
snapping / 24-bit depth buffer are implementation-defined: PARITY UNPINNED for this file — it restates standard
top-left-rule rasterisation with perspective-correct attributes, in float32 with the operation order of
csrc/render.hip, and is additionally cross-checked against an analytic ray-cast sphere in the tests.
"""
import numpy as np

f32 = np.float32


def project(vertices, pose, K):
    """(V,3) model points -> (u, v, Z) float32, unfused, same association as project_kernel."""
    P = pose.astype(f32)
    x, y, z = (vertices[:, i].astype(f32) for i in range(3))
    X = ((P[0, 0] * x + P[0, 1] * y) + P[0, 2] * z) + P[0, 3]
    Y = ((P[1, 0] * x + P[1, 1] * y) + P[1, 2] * z) + P[1, 3]
    Z = ((P[2, 0] * x + P[2, 1] * y) + P[2, 2] * z) + P[2, 3]
    K = K.astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K[0, 0] * X / Z + K[0, 2]
        v = K[1, 1] * Y / Z + K[1, 2]
    return u.astype(f32), v.astype(f32), Z.astype(f32)


def _edge(ax, ay, bx, by, px, py):
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax)


def _top_left(dx, dy):
    return (dy == 0 and dx > 0) or dy < 0


def _tex_bilinear(tex, u, v):
    TH, TW = tex.shape[:2]
    x = u * f32(TW) - f32(0.5)
    y = v * f32(TH) - f32(0.5)
    xf, yf = np.floor(x), np.floor(y)
    fx, fy = (x - xf).astype(f32), (y - yf).astype(f32)
    x0 = np.clip(xf.astype(np.int64), 0, TW - 1)
    x1 = np.clip(xf.astype(np.int64) + 1, 0, TW - 1)
    y0 = np.clip(yf.astype(np.int64), 0, TH - 1)
    y1 = np.clip(yf.astype(np.int64) + 1, 0, TH - 1)
    t00, t01, t10, t11 = tex[y0, x0], tex[y0, x1], tex[y1, x0], tex[y1, x1]
    top = t00 + (t01 - t00) * fx[..., None]
    bot = t10 + (t11 - t10) * fx[..., None]
    return (top + (bot - top) * fy[..., None]).astype(f32)


def _lit(col, mp, mn, pose, light_offset, intensity, ratio):
    """Fragment stage of lib/render_glumpy/render_py_light_modelnet_multi.py:36-80 in float32, operation order of
    csrc/render.hip: col (...,3) texture colour on 0..255, mp / mn (...,3) interpolated model-space position / normal."""
    P = np.asarray(pose, f32)
    sgn = np.array([1, -1, -1], f32)                     # yz_flip of _get_view_mtx (:202-208)
    pos = np.stack([sgn[r] * ((((P[r, 0] * mp[..., 0] + P[r, 1] * mp[..., 1]) + P[r, 2] * mp[..., 2])) + P[r, 3])
                    for r in range(3)], -1).astype(f32)
    nrm = np.stack([sgn[r] * ((P[r, 0] * mn[..., 0] + P[r, 1] * mn[..., 1]) + P[r, 2] * mn[..., 2]) for r in range(3)], -1).astype(f32)
    L = (np.asarray(light_offset, f32) + sgn * P[:, 3]).astype(f32)     # tester.py:161-165
    s2l = (L - pos).astype(f32)
    dotp = (nrm[..., 0] * s2l[..., 0] + nrm[..., 1] * s2l[..., 1]) + nrm[..., 2] * s2l[..., 2]
    ls = np.sqrt((s2l[..., 0] * s2l[..., 0] + s2l[..., 1] * s2l[..., 1]) + s2l[..., 2] * s2l[..., 2]).astype(f32)
    ln = np.sqrt((nrm[..., 0] * nrm[..., 0] + nrm[..., 1] * nrm[..., 1]) + nrm[..., 2] * nrm[..., 2]).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        br = (dotp / (ls * ln)).astype(f32)
    br = np.where(np.isnan(br), f32(0), np.maximum(np.minimum(br, f32(1)), f32(0))).astype(f32)
    shade = ((f32(1) - f32(ratio)) + f32(ratio) * br).astype(f32)
    inten = np.ones(3, f32) if intensity is None else np.asarray(intensity, f32)
    c = (col / f32(255)) * (shade[..., None] * inten)
    c = np.minimum(np.maximum(c, f32(0)), f32(1)).astype(f32)
    return np.rint(c * f32(255)).astype(f32)             # np.round(rgb * 255).astype(uint8), :165


def render(vertices, vertex_attr, faces, pose, K, H, W, znear=0.25, zfar=6.0, texture=None, pixel_means=None, normals=None,
           light_offset=None, light_intensity=None, brightness_ratio=0.7):
    """One pose of one mesh -> (image (3,H,W) RGB − means, depth (H,W)), float32. With `normals` (V,3) + `light_offset` the lit
    fragment stage of the ModelNet render machine (`_lit`)."""
    u, v, Z = project(np.asarray(vertices), np.asarray(pose), np.asarray(K))
    verts32 = np.asarray(vertices, dtype=f32)
    nrm32 = None if normals is None else np.asarray(normals, dtype=f32)
    attr = np.asarray(vertex_attr, dtype=f32)
    tex = None if texture is None else np.asarray(texture, dtype=f32)
    zbuf = np.full((H, W), np.inf, dtype=f32)
    rgb = np.zeros((H, W, 3), dtype=f32)
    for f in range(len(faces)):
        ia, ib, ic = (int(i) for i in faces[f])
        if not (Z[ia] > znear and Z[ib] > znear and Z[ic] > znear):
            continue
        area = _edge(u[ia], v[ia], u[ib], v[ib], u[ic], v[ic])
        if area == 0 or area != area:
            continue
        if area < 0:
            ib, ic = ic, ib
        ax, ay, az = u[ia], v[ia], Z[ia]
        bx, by, bz = u[ib], v[ib], Z[ib]
        cx, cy, cz = u[ic], v[ic], Z[ic]
        minx, maxx = min(ax, bx, cx), max(ax, bx, cx)
        miny, maxy = min(ay, by, cy), max(ay, by, cy)
        if not (maxx >= 0 and minx <= W - 1 and maxy >= 0 and miny <= H - 1):
            continue
        x0, x1 = max(0, int(np.ceil(minx))), min(W - 1, int(np.floor(maxx)))
        y0, y1 = max(0, int(np.ceil(miny))), min(H - 1, int(np.floor(maxy)))
        if x1 < x0 or y1 < y0:
            continue
        px, py = np.meshgrid(np.arange(x0, x1 + 1, dtype=f32), np.arange(y0, y1 + 1, dtype=f32))
        w0 = _edge(bx, by, cx, cy, px, py)
        w1 = _edge(cx, cy, ax, ay, px, py)
        w2 = _edge(ax, ay, bx, by, px, py)
        inside = ((w0 > 0) | ((w0 == 0) & _top_left(cx - bx, cy - by))) & \
                 ((w1 > 0) | ((w1 == 0) & _top_left(ax - cx, ay - cy))) & \
                 ((w2 > 0) | ((w2 == 0) & _top_left(bx - ax, by - ay)))
        if not inside.any():
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            q0, q1, q2 = w0 / az, w1 / bz, w2 / cz
            qs = (q0 + q1) + q2
            z = f32(1.0) / (qs / ((w0 + w1) + w2))
            if tex is None:
                col = ((q0[..., None] * attr[ia] + q1[..., None] * attr[ib]) + q2[..., None] * attr[ic]) / qs[..., None]
            else:
                tu = ((q0 * attr[ia, 0] + q1 * attr[ib, 0]) + q2 * attr[ic, 0]) / qs
                tv = ((q0 * attr[ia, 1] + q1 * attr[ib, 1]) + q2 * attr[ic, 1]) / qs
                col = _tex_bilinear(tex, tu.astype(f32), tv.astype(f32))
            if nrm32 is not None:
                mp = ((q0[..., None] * verts32[ia] + q1[..., None] * verts32[ib]) + q2[..., None] * verts32[ic]) / qs[..., None]
                mn = ((q0[..., None] * nrm32[ia] + q1[..., None] * nrm32[ib]) + q2[..., None] * nrm32[ic]) / qs[..., None]
                col = _lit(col.astype(f32), mp.astype(f32), mn.astype(f32), pose, light_offset, light_intensity, brightness_ratio)
        sub_z = zbuf[y0:y1 + 1, x0:x1 + 1]
        sub_c = rgb[y0:y1 + 1, x0:x1 + 1]
        win = inside & (z > znear) & (z < zfar) & (z < sub_z)   # strict <: first triangle wins exact ties
        sub_z[win] = z[win]
        sub_c[win] = col[win]
    depth = np.where(np.isinf(zbuf), f32(0), zbuf).astype(f32)
    means = np.zeros(3, f32) if pixel_means is None else np.asarray(pixel_means, f32)
    image = rgb.transpose(2, 0, 1) - means[:, None, None]
    return image.astype(f32), depth
