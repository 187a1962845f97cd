import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from oracle import net as onet
ctx = Context.get(0); cf = ctypes.c_float
def to_nc8(x):
    B, C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(B, C // 8, 8, H, W).transpose(0, 1, 3, 4, 2))
def from_nc8(y, shape):
    B, C, H, W = shape
    return np.ascontiguousarray(y.reshape(B, C // 8, H, W, 8).transpose(0, 1, 4, 2, 3).reshape(B, C, H, W))
for (B, cin, H, W, cout) in [(1, 8, 4, 2, 64), (1, 8, 8, 4, 64), (5, 8, 7, 9, 64), (1, 16, 8, 8, 64), (2, 256, 12, 16, 256)]:
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    want = onet.conv2d(x, w, b, 1, 1, 1.0)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_wino42_packed_size(cout, cin) // 4,))
    lib.deepim_conv_wino42_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin)
    out = ctx.array(np.full((B, cout, H, W), 7.0, np.float32))
    lib.deepim_conv2d_wino42_forward(ctx.handle, out, ctx.array(to_nc8(x)), pk, ctx.array(b), B, cin, H, W, cout, cf(1.0), 1, 0, 0)
    got = from_nc8(out.asnumpy(), (B, cout, H, W))
    nan = np.isnan(got)
    print((B, cin, H, W, cout), "nan frac %.3f" % nan.mean(), "untouched(7.0) frac %.3f" % (got == 7.0).mean())
    if nan.any():
        idx = np.argwhere(nan)
        print("  nan channels", sorted(set(idx[:, 1].tolist()))[:40], "rows", sorted(set(idx[:, 2].tolist())), "cols", sorted(set(idx[:, 3].tolist())))
    ok = ~nan
    d = np.abs(got - want)
    print("  max err over non-nan %.3e (scale %.2f)" % (d[ok].max() if ok.any() else -1, np.abs(want).max()))
    if not nan.any() and d.max() > 1e-4:
        e = np.argwhere(d > 1e-4)
        print("  bad channels", sorted(set(e[:, 1].tolist()))[:40], "rows", sorted(set(e[:, 2].tolist())), "cols", sorted(set(e[:, 3].tolist())))
        print("  got/want sample", got[0, 0, :, :2].ravel()[:8], want[0, 0, :, :2].ravel()[:8])
