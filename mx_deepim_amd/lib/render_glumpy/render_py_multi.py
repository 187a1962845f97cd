"""`Render_Py` with the constructor and `render()` signature of lib/render_glumpy/render_py_multi.py:21-129,
drawn by the HIP rasteriser (csrc/render.hip, `deepim_render_forward`) instead of an off-screen OpenGL window.

    render_machine = Render_Py(model_dir, classes, K, width, height, zNear, zFar)
    bgr, depth = render_machine.render(cls_idx, quat_or_mat, t)          # reference call (:101), host arrays
    image, depth = render_machine.render_batch(class_index, poses)       # device tensors for the batch updater

`render` returns what the reference returns (:121-129): (H,W,3) float32 BGR on the 0..255 scale and (H,W)
metric depth with 0 background. `render_batch` skips that host round trip: it writes the network-input tensors
(RGB − reversed PIXEL_MEANS, NCHW; batch_updater_py_multi.py:117-133) straight into HBM.

Meshes come from `<model_dir>/<class>/textured.obj` + `texture_map.png` like the reference (:68-77), or from
memory through `meshes={class: dict(vertices, faces, uv, texture | colors)}` (what the tests and bench use:
there are no model files offline).
"""
import ctypes
import os

import numpy as np

from ...runtime import Context, lib


def quat2mat(q):
    """(w,x,y,z) → 3x3 rotation, any non-zero norm (argument staging for `render`; the reference takes it from
    lib/pair_matching/RT_transform.py:quat2mat). A near-zero quaternion gives the identity, as there."""
    w, x, y, z = (float(c) for c in np.asarray(q, dtype=np.float64).reshape(4))
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / n
    return np.array([[1 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
                     [s * (x * y + w * z), 1 - s * (x * x + z * z), s * (y * z - w * x)],
                     [s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x * x + y * y)]])


def load_obj(path):
    """Minimal Wavefront reader for the `v` / `vt` / `f` records of the LINEMOD `textured.obj` files.
    Like glumpy's data.objload (used at render_py_multi.py:72) every distinct (position, texcoord) pair becomes
    one vertex; polygons are fanned into triangles."""
    pos, tex, corner, faces = [], [], {}, []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                pos.append([float(x) for x in tok[1:4]])
            elif tok[0] == "vt":
                tex.append([float(x) for x in tok[1:3]])
            elif tok[0] == "f":
                idx = []
                for c in tok[1:]:
                    parts = c.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    key = (vi - 1 if vi > 0 else len(pos) + vi, (ti - 1 if ti > 0 else len(tex) + ti) if ti else -1)
                    idx.append(corner.setdefault(key, len(corner)))
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    keys = sorted(corner, key=corner.get)
    vertices = np.array([pos[k[0]] for k in keys], dtype=np.float32).reshape(-1, 3)
    uv = np.array([tex[k[1]] if k[1] >= 0 else [0.0, 0.0] for k in keys], dtype=np.float32).reshape(-1, 2)
    return vertices, uv, np.array(faces, dtype=np.int32).reshape(-1, 3)


def load_texture(path):
    """(h,w,3) float32 on the 0..255 scale, flipped so row 0 is v = 0 (render_py_multi.py:76)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    return np.ascontiguousarray(img[::-1])


class _Mesh(object):
    def __init__(self, ctx, vertices, faces, uv=None, texture=None, colors=None):
        vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        if faces.size and (faces.min() < 0 or faces.max() >= len(vertices)):
            raise ValueError("face index out of range")
        self.V, self.F = len(vertices), len(faces)
        self.vertices = ctx.array(vertices)
        self.faces = ctx.array(faces, dtype=np.int32)
        if texture is not None:
            texture = np.ascontiguousarray(texture, dtype=np.float32)
            if uv is None or texture.ndim != 3 or texture.shape[2] != 3:
                raise ValueError("textured mesh needs uv (V,2) and texture (h,w,3)")
            self.attr = ctx.array(np.ascontiguousarray(uv, dtype=np.float32).reshape(self.V, 2))
            self.texture, self.tex_h, self.tex_w = ctx.array(texture), texture.shape[0], texture.shape[1]
        else:
            if colors is None:
                raise ValueError("mesh needs a texture or per-vertex colors")
            self.attr = ctx.array(np.ascontiguousarray(colors, dtype=np.float32).reshape(self.V, 3))
            self.texture, self.tex_h, self.tex_w = None, 0, 0


class Render_Py(object):
    def __init__(self, model_dir, classes, K, width=640, height=480, zNear=0.25, zFar=6.0, meshes=None, ctx=None,
                 pixel_means=None):
        self.width, self.height, self.zNear, self.zFar = int(width), int(height), float(zNear), float(zFar)
        self.K = np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        self.model_dir, self.classes = model_dir, list(classes)
        self.ctx = ctx or Context.default()
        # tensor-channel order (RGB): the updater subtracts PIXEL_MEANS[[2,1,0]] (batch_updater_py_multi.py:124-127)
        self.pixel_means = None if pixel_means is None else np.ascontiguousarray(pixel_means, np.float32).reshape(3)
        self.mesh_list = []
        for cls in self.classes:
            if meshes is not None and cls in meshes:
                self.mesh_list.append(_Mesh(self.ctx, **meshes[cls]))
                continue
            folder = os.path.join(model_dir, cls)
            vertices, uv, faces = load_obj(os.path.join(folder, "textured.obj"))
            self.mesh_list.append(_Mesh(self.ctx, vertices, faces, uv=uv,
                                        texture=load_texture(os.path.join(folder, "texture_map.png"))))

    # -- device API ----------------------------------------------------------------------------------------------
    def render_into(self, image, depth, cls_idx, poses, K=None, pixel_means="default", mask_rendered=None, mask_box=None,
                    mask_thresh=0.2, light_intensity=None):
        """image (n,3,H,W), depth (n,1,H,W), poses (n,3,4): all device arrays; one mesh. With `mask_rendered` (n,1,H,W)
        the same pass also writes depth > mask_thresh, and with `mask_box` its box_rendered rectangle. (`light_intensity` belongs
        to the lit ModelNet machine, render_py_light_modelnet_multi.py; this unlit draw ignores it.)"""
        m = self.mesh_list[int(cls_idx)]
        K = self.K if K is None else np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        means = self.pixel_means if isinstance(pixel_means, str) else pixel_means
        n = poses.shape[0]
        tail = (m.vertices, m.attr, m.faces, m.texture, m.tex_h, m.tex_w, poses, K, means, m.V, m.F, n, self.height,
                self.width, ctypes.c_float(self.zNear), ctypes.c_float(self.zFar))
        if mask_rendered is None:
            if mask_box is not None:
                raise ValueError("mask_box needs mask_rendered")
            lib.deepim_render_forward(self.ctx.handle, image, depth, *tail)
        else:
            lib.deepim_render_update_forward(self.ctx.handle, image, depth, mask_rendered, mask_box,
                                             ctypes.c_float(mask_thresh), *tail)

    def render_batch(self, class_index, poses, K=None, out=None, mask_rendered=None, mask_thresh=0.2, light_intensity=None):
        """poses (B,3,4) device; class_index: scalar, host sequence of B class ids, or None (= class 0).
        Samples of the same class in consecutive runs are drawn by one launch group. `out` = (image, depth) preallocated device
        tensors; `mask_rendered` (B,1,H,W): also written, = depth > mask_thresh, by the same pass."""
        B = poses.shape[0]
        image, depth = out if out is not None else (self.ctx.empty((B, 3, self.height, self.width)),
                                                    self.ctx.empty((B, 1, self.height, self.width)))
        if class_index is None:
            ids = np.zeros(B, np.int64)
        else:
            ids = class_index.asnumpy() if hasattr(class_index, "asnumpy") else np.asarray(class_index)
            ids = np.broadcast_to(ids.astype(np.int64).reshape(-1), (B,)) if ids.size == 1 else ids.astype(np.int64).reshape(B)
        b0 = 0
        while b0 < B:
            b1 = b0 + 1
            while b1 < B and ids[b1] == ids[b0]:
                b1 += 1
            self.render_into(image[b0:b1], depth[b0:b1], ids[b0], poses[b0:b1], K=K,
                             mask_rendered=None if mask_rendered is None else mask_rendered[b0:b1], mask_thresh=mask_thresh,
                             light_intensity=None if light_intensity is None else light_intensity[b0:b1])
            b0 = b1
        return image, depth

    # -- reference API -------------------------------------------------------------------------------------------
    def render(self, cls_idx, r, t, r_type="quat", K=None):
        if r_type == "quat":
            R = quat2mat(r)
        elif r_type == "mat":
            R = np.asarray(r)
        else:
            raise ValueError("r_type must be 'quat' or 'mat'")
        pose = np.zeros((1, 3, 4), np.float32)
        pose[0, :, :3] = np.asarray(R, np.float32).reshape(3, 3)
        pose[0, :, 3] = np.asarray(t, np.float32).reshape(3)
        image = self.ctx.empty((1, 3, self.height, self.width))
        depth = self.ctx.empty((1, 1, self.height, self.width))
        self.render_into(image, depth, cls_idx, self.ctx.array(pose), K=K, pixel_means=None)
        rgb = image.asnumpy()[0]
        bgr = np.ascontiguousarray(rgb[::-1].transpose(1, 2, 0))
        return bgr, depth.asnumpy()[0, 0]
