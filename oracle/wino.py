"""CPU restatement of the two algebraic identities csrc/wino.hip rests on (TEST INFRASTRUCTURE — only tests/ may import this).

The reference computes these layers with MXNet's Convolution (deepim/symbols/deepIM_flownet.py:65-101: conv2 / conv3 k5 s2 p2,
conv3_1 / conv4_1 / conv5_1 / conv6_1 k3 s1 p1); which algorithm cuDNN picks underneath is unspecified, and fp32 Winograd is one of
its choices. PARITY UNPINNED by the reference for this file: it states textbook identities (Lavin & Gray 2016, F(2x2,3x3); polyphase
decomposition of a strided correlation) in numpy, checked against the direct convolution of oracle/net.c.

  winograd_f2x2_3x3(x, w)      Y = A^T [ (G g G^T) . (B^T d B) ] A summed over input channels, per 2x2 output tile
  space_to_depth(x) / s2d_weights_5x5(w)   a k5 s2 p2 convolution == a k3 s1 p1 convolution of the 4-phase tensor
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd_f2x2_3x3(x, w, dtype=np.float64):
    """(B,Cin,H,W) * (Cout,Cin,3,3), stride 1, pad 1 -> (B,Cout,H,W); transforms and the channel sum carried in `dtype`
    (float32: the kernel's arithmetic, U rounded once from double as deepim_conv_wino_pack_weights does)."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    TY, TX = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((B, Cin, 2 * TY + 2, 2 * TX + 2), dtype)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    U = np.einsum("xa,ocab,nb->ocxn", G, w.astype(np.float64), G).astype(dtype)            # (Cout,Cin,4,4)
    bt = BT.astype(dtype)
    at = AT.astype(dtype)
    out = np.zeros((B, Cout, 2 * TY, 2 * TX), dtype)
    for ty in range(TY):
        for tx in range(TX):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                              # (B,Cin,4,4)
            V = np.einsum("xi,bcij,nj->bcxn", bt, d, bt).astype(dtype)
            M = np.einsum("ocxn,bcxn->boxn", U, V).astype(dtype)
            out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ax,boxn,en->boae", at, M, at)
    return out[:, :, :H, :W]


def space_to_depth(x):
    """(B,C,H,W), even H and W -> (B,4C,H/2,W/2) with channel (py*2+px)*C + c = x[:, c, py::2, px::2]."""
    return np.concatenate([x[:, :, py::2, px::2] for py in (0, 1) for px in (0, 1)], axis=1)


def s2d_weights_5x5(w):
    """(Cout,Cin,5,5) stride-2 pad-2 kernel -> (Cout,4Cin,3,3) stride-1 pad-1 kernel over space_to_depth(x):
    tap (a,b) of phase (py,px) = w[2a+py][2b+px], zero where the index reaches 5."""
    Cout, Cin = w.shape[:2]
    out = np.zeros((Cout, 4 * Cin, 3, 3), w.dtype)
    for py in (0, 1):
        for px in (0, 1):
            sub = w[:, :, py::2, px::2]                      # (3 or 2) x (3 or 2) taps
            out[:, (py * 2 + px) * Cin:(py * 2 + px + 1) * Cin, :sub.shape[2], :sub.shape[3]] = sub
    return out
