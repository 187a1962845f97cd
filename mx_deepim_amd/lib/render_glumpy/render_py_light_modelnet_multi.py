"""`Render_Py_Light_ModelNet_Multi` with the constructor and `render()` signature of
lib/render_glumpy/render_py_light_modelnet_multi.py:83-175 — the render machine of the ModelNet loops (BASELINE config 5:
deepim/core/tester.py:114-184, lib/pair_matching/batch_updater_py_multi.py:185-228) — drawn by the HIP rasteriser with the
per-fragment diffuse term of its shader (csrc/render.hip `deepim_render_lit_forward`) instead of an off-screen OpenGL window.

    rm = Render_Py_Light_ModelNet_Multi(model_path_list, texture_path, K, width, height, zNear, zFar, brightness_ratios=[0.7])
    bgr, depth = rm.render(model_idx, R_or_quat, t, light_position, light_intensity, brightness_k=0, r_type="mat")   # reference call
    image, depth = rm.render_batch(class_index, poses)          # device tensors, light set up as the reference's loops do

Meshes: `<model>.obj` (positions, texcoords, NORMALS; rescaled to the unit cube like glumpy's objload(rescale=True), then / 10,
:99-100) + one grey texture, or from memory through `meshes=[dict(vertices, faces, uv, normals, texture)]`.
"""
import ctypes

import numpy as np

from ...runtime import Context, lib
from .render_py_multi import Render_Py, _Mesh, load_texture, quat2mat

# light position of the reference's render closures (tester.py:146-160, batch_updater_py_multi.py:188-203): idx = 2 → [0, 1, 1] * 0.5
LIGHT_OFFSET = np.array([0.0, 1.0, 1.0], np.float32) * np.float32(0.5)


def load_obj_with_normals(path):
    """`v` / `vt` / `vn` / `f` records → (vertices, uv, normals, faces); every distinct (position, texcoord, normal) corner is one
    vertex (glumpy's data.objload, used at :99)."""
    pos, tex, nor, corner, faces = [], [], [], {}, []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                pos.append([float(x) for x in tok[1:4]])
            elif tok[0] == "vt":
                tex.append([float(x) for x in tok[1:3]])
            elif tok[0] == "vn":
                nor.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                idx = []
                for c in tok[1:]:
                    parts = (c.split("/") + ["", ""])[:3]
                    key = tuple(int(q) - 1 if q else -1 for q in parts)
                    idx.append(corner.setdefault(key, len(corner)))
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    keys = sorted(corner, key=corner.get)
    vertices = np.array([pos[k[0]] for k in keys], dtype=np.float32).reshape(-1, 3)
    uv = np.array([tex[k[1]] if k[1] >= 0 else [0.0, 0.0] for k in keys], dtype=np.float32).reshape(-1, 2)
    normals = np.array([nor[k[2]] if k[2] >= 0 else [0.0, 0.0, 0.0] for k in keys], dtype=np.float32).reshape(-1, 3)
    return vertices, uv, normals, np.array(faces, dtype=np.int32).reshape(-1, 3)


def rescale_unit(vertices):
    """glumpy objload(rescale=True): centre on the bounding box and scale its largest extent to [-1, 1]."""
    v = np.asarray(vertices, np.float32)
    lo, hi = v.min(0), v.max(0)
    return ((v - (lo + hi) / 2) / ((hi - lo).max() / 2)).astype(np.float32)


class Render_Py_Light_ModelNet_Multi(Render_Py):
    def __init__(self, model_path_list, texture_path, K, width=640, height=480, zNear=0.25, zFar=6.0, brightness_ratios=[0.7],
                 meshes=None, ctx=None, pixel_means=None):
        self.width, self.height, self.zNear, self.zFar = int(width), int(height), float(zNear), float(zFar)
        self.K = np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        self.model_path_list = list(model_path_list)
        self.classes = self.model_path_list
        self.brightness_ratios = [float(r) for r in brightness_ratios]
        self.ctx = ctx or Context.default()
        self.pixel_means = None if pixel_means is None else np.ascontiguousarray(pixel_means, np.float32).reshape(3)
        self.mesh_list, self.normal_list = [], []
        texture = None if meshes is not None else load_texture(texture_path)
        for i, path in enumerate(self.model_path_list):
            if meshes is not None:
                m = dict(meshes[i])
                normals = m.pop("normals")
            else:
                vertices, uv, normals, faces = load_obj_with_normals(path)
                m = dict(vertices=rescale_unit(vertices) / np.float32(10.0), faces=faces, uv=uv, texture=texture)   # :99-100
            self.mesh_list.append(_Mesh(self.ctx, **m))
            normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
            if len(normals) != self.mesh_list[-1].V:
                raise ValueError("one normal per vertex")
            self.normal_list.append(self.ctx.array(normals))

    # -- device API ----------------------------------------------------------------------------------------------
    def render_into(self, image, depth, cls_idx, poses, K=None, pixel_means="default", mask_rendered=None, mask_box=None,
                    mask_thresh=0.2, light_offset=None, light_intensity=None, brightness_k=0):
        """As Render_Py.render_into, lit: light at `light_offset` (default 0.5·[0,1,1]) + (t_x, −t_y, −t_z) of every sample's
        pose, `light_intensity` = device (n,3) or None (white, 1.0)."""
        m, nrm = self.mesh_list[int(cls_idx)], self.normal_list[int(cls_idx)]
        K = self.K if K is None else np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        means = self.pixel_means if isinstance(pixel_means, str) else pixel_means
        off = np.ascontiguousarray(LIGHT_OFFSET if light_offset is None else light_offset, np.float32).reshape(3)
        lib.deepim_render_lit_forward(self.ctx.handle, image, depth, mask_rendered, mask_box, ctypes.c_float(mask_thresh), m.vertices,
                                      m.attr, nrm, m.faces, m.texture, m.tex_h, m.tex_w, poses, K, means, off, light_intensity,
                                      ctypes.c_float(self.brightness_ratios[brightness_k]), m.V, m.F, poses.shape[0], self.height,
                                      self.width, ctypes.c_float(self.zNear), ctypes.c_float(self.zFar))

    # -- reference API -------------------------------------------------------------------------------------------
    def render(self, model_idx, r, t, light_position, light_intensity, brightness_k=0, r_type="quat"):
        """→ (bgr (H,W,3) uint8, depth (H,W)) as :138-175. `light_position` is the absolute position in OpenGL camera coordinates,
        as the reference's callers compute it (offset + (t_x, −t_y, −t_z))."""
        if r_type == "quat":
            R = quat2mat(r)
        elif r_type == "mat":
            R = np.asarray(r)
        else:
            raise ValueError("r_type must be 'quat' or 'mat'")
        t = np.asarray(t, np.float32).reshape(3)
        pose = np.zeros((1, 3, 4), np.float32)
        pose[0, :, :3] = np.asarray(R, np.float32).reshape(3, 3)
        pose[0, :, 3] = t
        off = np.asarray(light_position, np.float32).reshape(3) - np.array([t[0], -t[1], -t[2]], np.float32)
        image = self.ctx.empty((1, 3, self.height, self.width))
        depth = self.ctx.empty((1, 1, self.height, self.width))
        inten = self.ctx.array(np.asarray(light_intensity, np.float32).reshape(1, 3))
        self.render_into(image, depth, model_idx, self.ctx.array(pose), pixel_means=None, light_offset=off, light_intensity=inten,
                         brightness_k=brightness_k)
        rgb = image.asnumpy()[0]
        bgr = np.ascontiguousarray(rgb[::-1].transpose(1, 2, 0)).astype(np.uint8)
        return bgr, depth.asnumpy()[0, 0]
