import ctypes, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from oracle import net as onet
ctx = Context.get(0)
cf = ctypes.c_float
def nc8(x):
    B, C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(B, C // 8, 8, H, W).transpose(0, 1, 3, 4, 2))
def from_nc8(y, C):
    B, C8, H, W, _ = y.shape
    return np.ascontiguousarray(y.transpose(0, 1, 4, 2, 3).reshape(B, C, H, W))
rng = np.random.default_rng(0)
for (B, cin, H, W, cout, k, s, p) in [(2, 64, 60, 80, 128, 5, 2, 2), (1, 128, 30, 40, 256, 3, 1, 1), (3, 16, 17, 23, 72, 3, 1, 1), (1, 512, 8, 10, 1024, 3, 2, 1)]:
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, k, k) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin, k, k)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ref2 = onet.conv2d(x, w, b, s, p, 0.1, pair_order=2)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    xin = ctx.array(nc8(x))
    for ms in (1, 0):
        lib.deepim_set_option(ctx.handle, b"conv_max_split", ms)
        for onc8 in (1, 0):
            if onc8 and cout % 8: continue
            out = ctx.zeros((B, cout, Ho, Wo))
            lib.deepim_conv2d_forward_ex(ctx.handle, out, xin, pk, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, 1, onc8)
            got = out.asnumpy()
            if onc8:
                got = from_nc8(got.reshape(B, cout // 8, Ho, Wo, 8), cout)
            print((B, cin, H, W, cout, k, s, p), "split" if not ms else "nosplit", "out_nc8" if onc8 else "out_nchw",
                  "bit-exact vs order-2 oracle:", np.array_equal(got, ref2), " rel err vs canonical %.1e" % (np.abs(got - ref).max() / np.abs(ref).max()))
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    # relayout kernel
    d = ctx.empty(x.shape)
    lib.deepim_relayout_nc8(ctx.handle, d, ctx.array(x), B, cin, H * W, 1)
    assert np.array_equal(d.asnumpy().reshape(B, cin // 8, H, W, 8), nc8(x))
    e = ctx.empty(x.shape)
    lib.deepim_relayout_nc8(ctx.handle, e, d, B, cin, H * W, 0)
    assert np.array_equal(e.asnumpy(), x)
print("relayout ok")
