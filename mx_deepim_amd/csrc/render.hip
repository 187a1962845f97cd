// GPU rasteriser for the step BETWEEN refinement iterations: replaces the OpenGL off-screen
// render + glReadPixels of lib/render_glumpy/render_py_multi.py:101-129 so the render-and-compare loop never
// leaves the device. Same camera model: u0 = cx + 0.5 with GL pixel centres at +0.5 (:132-147) means a camera
// point projects to pixel-INDEX coordinates u = fx·X/Z + cx, v = fy·Y/Z + cy and a pixel is covered when its
// index (i, j) lies inside the projected triangle; depth test GL_LESS, no face culling, no blending (:93-95);
// depth read-back is linearised to metric z (:126-128) — here z is kept metric throughout; colour is the
// texture (GL_LINEAR, clamp) at perspective-correct uv, or perspective-correct vertex colours.
// OpenGL's exact rasterisation (sub-pixel snapping, 24-bit depth) cannot run here: PARITY UNPINNED, restated as
// standard top-left-rule rasterisation; the tests check it against a CPU restatement and an analytic ray cast.
//
// Three kernels, all per-pair batched:
//   project:  one thread per (pair, vertex): camera transform + projection → (u, v, Z)
//   raster:   one thread per (pair, triangle): walks the triangle's clipped bounding box, interpolates 1/Z
//             (affine in screen space, like GL's z) and resolves visibility with ONE 64-bit atomicMin per
//             covered pixel on the key (float_bits(Z) << 32 | triangle id) — first triangle wins exact ties
//   resolve:  one thread per pixel: re-derives the barycentrics of the winning triangle, shades, writes the
//             network tensors directly: image (B,3,H,W) RGB mean-subtracted, depth (B,1,H,W), 0 = background
#include "common.h"
#include <limits.h>

namespace {

struct PV { float u, v, z; };

__global__ __launch_bounds__(256) void project_kernel(PV* __restrict__ pv, const float* __restrict__ verts,
                                                      const float* __restrict__ poses, Mat3 K, int V,
                                                      int* __restrict__ box_words) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  // arm this sample's bbox accumulator for the resolve pass of THIS call, in stream order (two launches earlier):
  // no state survives between calls, so changing B between calls or replaying a captured graph is safe
  if (box_words != nullptr && i < 4) box_words[b * 4 + i] = (i & 1) ? -1 : INT_MAX;
  if (i >= V) return;
  const float* P = poses + b * 12;
  const float x = verts[i * 3], y = verts[i * 3 + 1], z = verts[i * 3 + 2];
  const float X = ((P[0] * x + P[1] * y) + P[2] * z) + P[3];
  const float Y = ((P[4] * x + P[5] * y) + P[6] * z) + P[7];
  const float Z = ((P[8] * x + P[9] * y) + P[10] * z) + P[11];
  PV o;
  o.u = K.v[0] * X / Z + K.v[2];
  o.v = K.v[4] * Y / Z + K.v[5];
  o.z = Z;
  pv[(long)b * V + i] = o;
}

// edge function of pixel (px,py) against edge a→b; > 0 on the interior side for the orientation used below
__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
  return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
// top-left rule (image coordinates, y down) for an edge with direction (dx,dy) of a triangle oriented so that edge_fn
// is positive inside: the top edge runs left→right, left edges run upwards
__device__ __forceinline__ bool top_left(float dx, float dy) { return (dy == 0.f && dx > 0.f) || dy < 0.f; }

struct Tri { float ax, ay, az, bx, by, bz, cx, cy, cz; bool ok; };

__device__ __forceinline__ Tri load_tri(const PV* __restrict__ pv, const int* __restrict__ faces, int f, float znear,
                                        int& ia, int& ib, int& ic) {
  ia = faces[f * 3]; ib = faces[f * 3 + 1]; ic = faces[f * 3 + 2];
  PV a = pv[ia], b = pv[ib], c = pv[ic];
  Tri t;
  t.ok = a.z > znear && b.z > znear && c.z > znear;  // triangles crossing the near plane are dropped, not clipped
  const float area = edge_fn(a.u, a.v, b.u, b.v, c.u, c.v);
  if (area == 0.f || !(area == area)) t.ok = false;
  if (area < 0.f) { PV tmp = b; b = c; c = tmp; const int ti = ib; ib = ic; ic = ti; }  // no culling: orient positively
  t.ax = a.u; t.ay = a.v; t.az = a.z; t.bx = b.u; t.by = b.v; t.bz = b.z; t.cx = c.u; t.cy = c.v; t.cz = c.z;
  return t;
}

// barycentric weights of pixel (x,y) (w0 ↔ a, w1 ↔ b, w2 ↔ c) and coverage under the top-left rule
__device__ __forceinline__ bool cover(const Tri& t, float x, float y, float& w0, float& w1, float& w2) {
  w0 = edge_fn(t.bx, t.by, t.cx, t.cy, x, y);
  w1 = edge_fn(t.cx, t.cy, t.ax, t.ay, x, y);
  w2 = edge_fn(t.ax, t.ay, t.bx, t.by, x, y);
  const bool i0 = w0 > 0.f || (w0 == 0.f && top_left(t.cx - t.bx, t.cy - t.by));
  const bool i1 = w1 > 0.f || (w1 == 0.f && top_left(t.ax - t.cx, t.ay - t.cy));
  const bool i2 = w2 > 0.f || (w2 == 0.f && top_left(t.bx - t.ax, t.by - t.ay));
  return i0 && i1 && i2;
}

__device__ __forceinline__ float pixel_depth(const Tri& t, float w0, float w1, float w2) {
  const float sum = (w0 + w1) + w2;
  const float inv = ((w0 / t.az + w1 / t.bz) + w2 / t.cz) / sum;  // 1/Z is affine in screen space
  return 1.0f / inv;
}

__global__ __launch_bounds__(256) void raster_kernel(unsigned long long* __restrict__ zbuf, const PV* __restrict__ pv_all,
                                                     const int* __restrict__ faces, int V, int F, int H, int W,
                                                     float znear, float zfar) {
  const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  int ia, ib, ic;
  const Tri t = load_tri(pv_all + (long)b * V, faces, f, znear, ia, ib, ic);
  if (!t.ok) return;
  const float minx = fminf(t.ax, fminf(t.bx, t.cx)), maxx = fmaxf(t.ax, fmaxf(t.bx, t.cx));
  const float miny = fminf(t.ay, fminf(t.by, t.cy)), maxy = fmaxf(t.ay, fmaxf(t.by, t.cy));
  if (!(maxx >= 0.f && minx <= (float)(W - 1) && maxy >= 0.f && miny <= (float)(H - 1))) return;
  const int x0 = max(0, (int)ceilf(minx)), x1 = min(W - 1, (int)floorf(maxx));
  const int y0 = max(0, (int)ceilf(miny)), y1 = min(H - 1, (int)floorf(maxy));
  unsigned long long* zb = zbuf + (long)b * H * W;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      float w0, w1, w2;
      if (!cover(t, (float)x, (float)y, w0, w1, w2)) continue;
      const float z = pixel_depth(t, w0, w1, w2);
      if (!(z > znear && z < zfar)) continue;  // GL clips fragments outside [zNear, zFar]
      const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)f;
      atomicMin(zb + (long)y * W + x, key);
    }
}

__device__ __forceinline__ float tex_bilinear(const float* __restrict__ tex, int TH, int TW, int c, float u, float v) {
  // GL_LINEAR with clamp-to-edge; texel centres at (i + 0.5) / size; v = 0 is the FIRST row of `tex`
  const float x = u * TW - 0.5f, y = v * TH - 0.5f;
  const float xf = floorf(x), yf = floorf(y);
  const float fx = x - xf, fy = y - yf;
  const int x0 = min(max((int)xf, 0), TW - 1), x1 = min(max((int)xf + 1, 0), TW - 1);
  const int y0 = min(max((int)yf, 0), TH - 1), y1 = min(max((int)yf + 1, 0), TH - 1);
  const float t00 = tex[(y0 * TW + x0) * 3 + c], t01 = tex[(y0 * TW + x1) * 3 + c];
  const float t10 = tex[(y1 * TW + x0) * 3 + c], t11 = tex[(y1 * TW + x1) * 3 + c];
  const float top = t00 + (t01 - t00) * fx, bot = t10 + (t11 - t10) * fx;
  return top + (bot - top) * fy;
}

// Lit fragment stage of the ModelNet loop (lib/render_glumpy/render_py_light_modelnet_multi.py:36-80, light set-up
// deepim/core/tester.py:146-172 = batch_updater_py_multi.py:185-228). All of it in OpenGL camera coordinates (y, z flipped):
//   position = u_view·u_model·v_position,  normal = (u_view·u_model)^-T·(v_normal, 1) — whose 4-vector normalisation cancels in
//   the next line —  brightness = clamp(n·(L − position) / (|L − position|·|n|), 0, 1),
//   colour = texture·((1 − r) + r·brightness)·intensity, read back as round(clamp(colour, 0, 1)·255) (:162-166: uint8).
// L = light_offset + (t_x, −t_y, −t_z) of the sample's pose. verts == nullptr: the unlit stage of render_py_multi.py.
struct LitParams {
  const float* verts;      // (V,3) model-space positions (v_position)
  const float* normals;    // (V,3) per-vertex normals (v_normal)
  const float* poses;      // (B,3,4)
  const float* intensity;  // (B,3) device, or nullptr = (1,1,1)
  Vec3 light_offset;       // 0.5·(0,1,1) in the reference's loops
  float ratio;             // brightness_ratio (0.7)
};

// attr: per-vertex attributes — 3 floats RGB (0..255) when tex == nullptr, else 2 floats uv
__global__ __launch_bounds__(256) void resolve_kernel(float* __restrict__ image, float* __restrict__ depth,
                                                      unsigned long long* __restrict__ zbuf,
                                                      const PV* __restrict__ pv_all, const int* __restrict__ faces,
                                                      const float* __restrict__ attr, const float* __restrict__ tex,
                                                      int TH, int TW, Vec3 means, int V, int H, int W, float znear,
                                                      float* __restrict__ mask, float mask_thresh,
                                                      int* __restrict__ box_words, LitParams lit) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const long plane = (long)H * W;
  const bool inside = p < plane;
  const unsigned long long key = inside ? zbuf[(long)b * plane + p] : ~0ull;
  if (inside && key != ~0ull) zbuf[(long)b * plane + p] = ~0ull;      // the z-buffer leaves the call as it entered it: all-ones
  float rgb[3] = {0.f, 0.f, 0.f};
  float z = 0.f;
  if (key != ~0ull) {
    const int f = (int)(unsigned)(key & 0xffffffffu);
    z = __uint_as_float((unsigned)(key >> 32));
    int ia, ib, ic;
    const Tri t = load_tri(pv_all + (long)b * V, faces, f, znear, ia, ib, ic);
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    float w0, w1, w2;
    cover(t, (float)x, (float)y, w0, w1, w2);
    // perspective-correct weights
    const float q0 = w0 / t.az, q1 = w1 / t.bz, q2 = w2 / t.cz;
    const float qs = (q0 + q1) + q2;
    if (tex == nullptr) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rgb[c] = ((q0 * attr[ia * 3 + c] + q1 * attr[ib * 3 + c]) + q2 * attr[ic * 3 + c]) / qs;
    } else {
      const float u = ((q0 * attr[ia * 2] + q1 * attr[ib * 2]) + q2 * attr[ic * 2]) / qs;
      const float v = ((q0 * attr[ia * 2 + 1] + q1 * attr[ib * 2 + 1]) + q2 * attr[ic * 2 + 1]) / qs;
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = tex_bilinear(tex, TH, TW, c, u, v);
    }
    if (lit.verts != nullptr) {
      const float* P = lit.poses + b * 12;
      float mp[3], mn[3];      // perspective-correct model-space position and normal of the fragment (GL varyings)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mp[c] = ((q0 * lit.verts[ia * 3 + c] + q1 * lit.verts[ib * 3 + c]) + q2 * lit.verts[ic * 3 + c]) / qs;
        mn[c] = ((q0 * lit.normals[ia * 3 + c] + q1 * lit.normals[ib * 3 + c]) + q2 * lit.normals[ic * 3 + c]) / qs;
      }
      float pos[3], nrm[3], s2l[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float sgn = r == 0 ? 1.f : -1.f;     // yz_flip of _get_view_mtx
        pos[r] = sgn * ((((P[r * 4] * mp[0] + P[r * 4 + 1] * mp[1]) + P[r * 4 + 2] * mp[2])) + P[r * 4 + 3]);
        nrm[r] = sgn * ((P[r * 4] * mn[0] + P[r * 4 + 1] * mn[1]) + P[r * 4 + 2] * mn[2]);
        s2l[r] = (lit.light_offset.v[r] + sgn * P[r * 4 + 3]) - pos[r];
      }
      const float dotp = (nrm[0] * s2l[0] + nrm[1] * s2l[1]) + nrm[2] * s2l[2];
      const float ls = sqrtf((s2l[0] * s2l[0] + s2l[1] * s2l[1]) + s2l[2] * s2l[2]);
      const float ln = sqrtf((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
      float br = dotp / (ls * ln);
      br = fmaxf(fminf(br, 1.f), 0.f);             // max(min(x, 1), 0): a NaN (degenerate normal) becomes 0, as in GLSL on most drivers
      if (!(br == br)) br = 0.f;
      const float shade = (1.f - lit.ratio) + lit.ratio * br;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float inten = lit.intensity ? lit.intensity[b * 3 + c] : 1.f;
        float col = (rgb[c] / 255.f) * (shade * inten);
        col = fminf(fmaxf(col, 0.f), 1.f);
        rgb[c] = rintf(col * 255.f);               // np.round(rgb * 255).astype(uint8)
      }
    }
  }
  if (inside) {
#pragma unroll
    for (int c = 0; c < 3; ++c) image[((long)b * 3 + c) * plane + p] = rgb[c] - means.v[c];
    depth[(long)b * plane + p] = z;
  }
  if (mask != nullptr) {
    // fused tester.py:440-442 (mask_rendered = depth > thresh) and the bbox pass of the box_rendered rectangle
    const bool on = inside && z > mask_thresh;
    if (inside) mask[(long)b * plane + p] = on ? 1.f : 0.f;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    int xmin = on ? x : INT_MAX, xmax = on ? x : -1, ymin = on ? y : INT_MAX, ymax = on ? y : -1;
    if (box_words != nullptr && __ballot(on) != 0ull) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        xmin = min(xmin, __shfl_xor(xmin, off, 64)); xmax = max(xmax, __shfl_xor(xmax, off, 64));
        ymin = min(ymin, __shfl_xor(ymin, off, 64)); ymax = max(ymax, __shfl_xor(ymax, off, 64));
      }
      if ((threadIdx.x & 63) == 0) {
        atomicMin(&box_words[b * 4 + 0], xmin); atomicMax(&box_words[b * 4 + 1], xmax);
        atomicMin(&box_words[b * 4 + 2], ymin); atomicMax(&box_words[b * 4 + 3], ymax);
      }
    }
  }
}

}  // namespace

static int render_impl(deepim_ctx* ctx, float* image, float* depth, float* mask, float* mask_box, float mask_thresh,
                       const float* vertices, const float* vertex_attr, const int32_t* faces, const float* texture,
                       int tex_h, int tex_w, const float* poses, const float* K_host, const float* pixel_means_host,
                       int V, int F, int B, int H, int W, float znear, float zfar, const float* normals = nullptr,
                       const float* light_offset_host = nullptr, const float* light_intensity = nullptr, float ratio = 0.f) {
  if (B == 0) return 0;
  DI_REQUIRE(V > 0 && F > 0 && H > 0 && W > 0, "render: empty mesh or image");
  DI_REQUIRE(znear > 0.f && zfar > znear, "render: need 0 < zNear < zFar");
  DI_REQUIRE(mask_box == nullptr || (mask != nullptr && B <= DI_MAX_BOX_SAMPLES), "render: mask_box needs mask, B <= 4096");
  const size_t zbytes = (size_t)B * H * W * sizeof(unsigned long long);
  const size_t pbytes = (size_t)B * V * sizeof(PV);
  void* scratch;
  int rc = deepim_scratch(ctx, pbytes + 64, &scratch);
  if (rc) return rc;
  // The z-buffer is the context's own (not the shared scratch): all-ones between calls — the resolve pass resets what the raster
  // pass touched — so a draw carries no clearing pass (8 B/px of writes and one launch per refinement iteration). Grow-only.
  if (ctx->zbuf_bytes < zbytes) {
    DI_REQUIRE(!ctx->capturing, "render: z-buffer growth during graph capture; run the sequence once eagerly first");
    DI_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->zbuf) DI_CHECK(hipFree(ctx->zbuf));
    ctx->zbuf = nullptr; ctx->zbuf_bytes = 0;
    DI_CHECK(hipMalloc((void**)&ctx->zbuf, zbytes));
    ctx->zbuf_bytes = zbytes;
    DI_CHECK(hipMemsetAsync(ctx->zbuf, 0xff, zbytes, ctx->stream));
    ctx->zbuf_dirty = 0;
  }
  if (ctx->zbuf_dirty) {      // an earlier draw rasterised but its resolve pass was never launched: stale depth keys
    DI_CHECK(hipMemsetAsync(ctx->zbuf, 0xff, ctx->zbuf_bytes, ctx->stream));
    ctx->zbuf_dirty = 0;
  }
  unsigned long long* zbuf = ctx->zbuf;
  PV* pv = (PV*)scratch;
  Mat3 K;
  for (int i = 0; i < 9; ++i) K.v[i] = K_host[i];
  Vec3 means = {{0, 0, 0}};
  if (pixel_means_host) for (int i = 0; i < 3; ++i) means.v[i] = pixel_means_host[i];
  int* words = mask_box ? ctx->box_words : nullptr;
  LitParams lit = {nullptr, nullptr, poses, light_intensity, {{0, 0, 0}}, ratio};
  if (normals != nullptr) {
    lit.verts = vertices; lit.normals = normals;
    if (light_offset_host) for (int i = 0; i < 3; ++i) lit.light_offset.v[i] = light_offset_host[i];
  }
  hipLaunchKernelGGL(project_kernel, dim3(di_div_up(V, 256), B), dim3(256), 0, ctx->stream, pv, vertices, poses, K, V,
                     words);
  DI_LAUNCH_CHECK();
  ctx->zbuf_dirty = 1;
  hipLaunchKernelGGL(raster_kernel, dim3(di_div_up(F, 256), B), dim3(256), 0, ctx->stream, zbuf, pv, (const int*)faces, V,
                     F, H, W, znear, zfar);
  hipLaunchKernelGGL(resolve_kernel, dim3(di_div_up((long)H * W, 256), B), dim3(256), 0, ctx->stream, image, depth, zbuf,
                     pv, (const int*)faces, vertex_attr, texture, tex_h, tex_w, means, V, H, W, znear, mask, mask_thresh,
                     words, lit);
  DI_LAUNCH_CHECK();
  ctx->zbuf_dirty = 0;        // the resolve pass is queued behind the raster pass: the buffer will be all-ones again
  if (mask_box) return deepim_mask_box_fill(ctx, mask_box, words, B, H, W);
  return 0;
}

extern "C" int deepim_render_forward(deepim_ctx* ctx, float* image, float* depth, const float* vertices,
                                     const float* vertex_attr, const int32_t* faces, const float* texture,
                                     int tex_h, int tex_w, const float* poses, const float* K_host,
                                     const float* pixel_means_host, int V, int F, int B, int H, int W, float znear,
                                     float zfar) {
  DI_DEVICE(ctx);
  return render_impl(ctx, image, depth, nullptr, nullptr, 0.f, vertices, vertex_attr, faces, texture, tex_h, tex_w, poses,
                     K_host, pixel_means_host, V, F, B, H, W, znear, zfar);
}

extern "C" int deepim_render_update_forward(deepim_ctx* ctx, float* image, float* depth, float* mask_rendered,
                                            float* mask_box, float mask_thresh, const float* vertices,
                                            const float* vertex_attr, const int32_t* faces, const float* texture,
                                            int tex_h, int tex_w, const float* poses, const float* K_host,
                                            const float* pixel_means_host, int V, int F, int B, int H, int W,
                                            float znear, float zfar) {
  DI_DEVICE(ctx);
  DI_REQUIRE(mask_rendered != nullptr, "render_update: mask_rendered is NULL");
  return render_impl(ctx, image, depth, mask_rendered, mask_box, mask_thresh, vertices, vertex_attr, faces, texture, tex_h,
                     tex_w, poses, K_host, pixel_means_host, V, F, B, H, W, znear, zfar);
}

// The same passes with the lit fragment stage of the ModelNet render machine (render_py_light_modelnet_multi.py:36-80,170-188;
// the render closures of tester.py:146-184 and batch_updater_py_multi.py:185-228): see LitParams above.
extern "C" int deepim_render_lit_forward(deepim_ctx* ctx, float* image, float* depth, float* mask_rendered, float* mask_box,
                                         float mask_thresh, const float* vertices, const float* vertex_attr,
                                         const float* normals, const int32_t* faces, const float* texture, int tex_h,
                                         int tex_w, const float* poses, const float* K_host, const float* pixel_means_host,
                                         const float* light_offset_host, const float* light_intensity,
                                         float brightness_ratio, int V, int F, int B, int H, int W, float znear, float zfar) {
  DI_DEVICE(ctx);
  DI_REQUIRE(normals != nullptr && light_offset_host != nullptr, "render_lit: normals and light offset are required");
  DI_REQUIRE(brightness_ratio >= 0.f && brightness_ratio <= 1.f, "render_lit: brightness_ratio in [0, 1]");
  DI_REQUIRE(mask_box == nullptr || mask_rendered != nullptr, "render_lit: mask_box needs mask_rendered");
  return render_impl(ctx, image, depth, mask_rendered, mask_box, mask_thresh, vertices, vertex_attr, faces, texture, tex_h, tex_w,
                     poses, K_host, pixel_means_host, V, F, B, H, W, znear, zfar, normals, light_offset_host, light_intensity,
                     brightness_ratio);
}
