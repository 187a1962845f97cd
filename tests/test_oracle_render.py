"""CPU checks of the re-render restatement (oracle/render.py) and of the host-side mesh readers."""
import numpy as np

from oracle import render as orender
from mx_deepim_amd import synthetic

K = synthetic.K_LINEMOD
AXES = [0.05, 0.04, 0.035]


def test_oracle_render_against_analytic_raycast():
    # the GL reference cannot run offline; pin the rasteriser restatement on geometry with a closed form instead
    mesh = synthetic.ellipsoid_mesh(AXES, 48, 96)
    rng = np.random.default_rng(3)
    pose = synthetic.sample_pose_pair(rng)[0]
    img, dep = orender.render(mesh["vertices"], mesh["colors"], mesh["faces"], pose, K, 480, 640)
    _, ref = synthetic.raycast_ellipsoid(pose, AXES)
    both = (dep > 0) & (ref > 0)
    assert both.sum() > 1000
    assert ((dep > 0) != (ref > 0)).sum() <= 0.03 * (ref > 0).sum()
    assert np.abs(dep - ref)[both].max() < 3e-3
    assert np.median(np.abs(dep - ref)[both]) < 2e-4
    assert img.shape == (3, 480, 640) and img.min() >= 0 and img.max() <= 255


def test_oracle_render_projection_convention():
    # render_py_multi.py:132-147: u0 = cx + 0.5 against GL's half-pixel centres ⇒ a point at pixel index (i, j) under K
    # lands on pixel (i, j). A tiny triangle around the back-projection of pixel (37, 21) must cover exactly that pixel.
    z, i, j = 1.5, 37, 21
    Kf = K.astype(np.float64)
    c = np.array([(i - Kf[0, 2]) * z / Kf[0, 0], (j - Kf[1, 2]) * z / Kf[1, 1], z])
    e = 0.4 * z / Kf[0, 0]
    verts = np.array([c + [-e, -e, 0], c + [e, -e, 0], c + [0, e, 0]], np.float32)
    img, dep = orender.render(verts, np.full((3, 3), 255, np.float32), np.array([[0, 1, 2]]), np.eye(4)[:3], K, 48, 64)
    ys, xs = np.nonzero(dep)
    assert list(zip(xs, ys)) == [(i, j)]
    assert abs(dep[j, i] - z) < 1e-6


def test_texture_sampling_is_gl_linear():
    tex = np.zeros((2, 2, 3), np.float32)
    tex[0, 0], tex[0, 1], tex[1, 0], tex[1, 1] = 0, 100, 200, 300
    u = np.array([0.25, 0.75, 0.5, 0.0, 1.0], np.float32)
    v = np.array([0.25, 0.25, 0.5, 0.0, 1.0], np.float32)
    out = orender._tex_bilinear(tex, u, v)[:, 0]
    np.testing.assert_allclose(out, [0, 100, 150, 0, 300])   # texel centres, mid-point blend, clamp at the border


def test_obj_and_texture_readers(tmp_path):
    from mx_deepim_amd.lib.render_glumpy.render_py_multi import load_obj, load_texture
    obj = tmp_path / "textured.obj"
    obj.write_text("# quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                   "f 1/1 2/2 3/3 4/4\nf 1/5 3/3 4/4\n")
    vertices, uv, faces = load_obj(str(obj))
    assert vertices.shape == (5, 3) and uv.shape == (5, 2)          # position 1 appears with two texcoords
    assert faces.tolist() == [[0, 1, 2], [0, 2, 3], [4, 2, 3]]
    np.testing.assert_array_equal(vertices[4], vertices[0])
    np.testing.assert_array_equal(uv[4], [0.5, 0.5])
    from PIL import Image
    arr = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    Image.fromarray(arr).save(str(tmp_path / "texture_map.png"))
    tex = load_texture(str(tmp_path / "texture_map.png"))
    np.testing.assert_array_equal(tex, arr[::-1].astype(np.float32))


def test_lit_fragment_stage_against_the_closed_form_on_a_sphere():
    """render_py_light_modelnet_multi.py:36-80 restated in oracle/render.py: on a sphere the interpolated normal is the radial
    direction, so every covered pixel's grey level has a closed form — texture·((1 − r) + r·max(0, n·(L − p)/|L − p|))·intensity
    with p the ray/sphere hit point, L = 0.5·(0,1,1) + (t_x, −t_y, −t_z) in OpenGL camera coordinates (tester.py:146-165)."""
    R_ = 0.05
    mesh = synthetic.ellipsoid_mesh([R_, R_, R_], 96, 192)
    normals = mesh["vertices"] / np.linalg.norm(mesh["vertices"], axis=1, keepdims=True)
    tex = np.full((4, 4, 3), 220.0, np.float32)
    t = np.array([0.03, -0.02, 0.55], np.float32)
    a = 0.4
    Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    pose = np.concatenate([Rm, t[:, None]], 1).astype(np.float32)
    inten = np.array([1.05, 0.95, 1.0], np.float32)
    img, dep = orender.render(mesh["vertices"], mesh["uv"], mesh["faces"], pose, K, 480, 640, texture=tex, normals=normals,
                              light_offset=[0.0, 0.5, 0.5], light_intensity=inten, brightness_ratio=0.7)
    ys, xs = np.nonzero(dep > 0)
    assert len(ys) > 3000
    Kd = K.astype(np.float64)
    ray = np.stack([(xs - Kd[0, 2]) / Kd[0, 0], (ys - Kd[1, 2]) / Kd[1, 1], np.ones(len(xs))], -1)
    p_cv = ray * dep[ys, xs][:, None]                                   # camera-space hit point (OpenCV axes)
    flip = np.array([1.0, -1.0, -1.0])
    p_gl, c_gl = p_cv * flip, t.astype(np.float64) * flip
    n_gl = (p_gl - c_gl) / np.linalg.norm(p_gl - c_gl, axis=1, keepdims=True)
    L = np.array([0.0, 0.5, 0.5]) + c_gl
    s2l = L - p_gl
    br = np.clip((n_gl * s2l).sum(1) / np.linalg.norm(s2l, axis=1), 0, 1)
    want = 220.0 * ((0.3 + 0.7 * br)[:, None] * inten[None])
    got = img[:, ys, xs].T
    inner = np.linalg.norm(p_gl - c_gl, axis=1) > 0                      # all
    err = np.abs(got - np.clip(want, 0, 255))
    assert np.median(err) <= 1.0 and np.percentile(err, 99) <= 4.0, (np.median(err), err.max())   # tessellated silhouette + rounding
    assert got.min() >= 0 and got.max() <= 255 and np.all(got == np.rint(got))                    # uint8 read-back
    assert br.max() > 0.95 and br.min() == 0.0                                                    # lit and unlit side both in view
