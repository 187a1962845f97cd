"""FlowNetS-backboned matching network of DeepIM as a resident device pipeline.

Mirrors ``deepim/symbols/deepIM_flownet.py`` (reference): ``get_symbol`` / ``get_test_symbol_share``
(:548-735), ``get_convs`` (:32-169) and ``init_weights`` (:753-845) keep their names and the MXNet
parameter naming (``flow_conv1_weight`` ... ``trans_bias``), but instead of an MXNet Symbol the
"symbol" is a bound chain of HIP launches on one stream:

    zoom (ZoomMask + ZoomImageWithFactor [+ ZoomDepth] + /255 + Concat, one fused front end)
    → 10 × (Convolution + bias + LeakyReLU 0.1) on the fp32 matrix cores
    → fc6 → fc7 → rot/trans + inverse ZoomTrans → se3
    [→ FlowNetS refinement decoder → mask / flow heads when the config keeps them in the test graph]
    → RT_transform (pose update, lib/pair_matching/RT_transform.py:127-151)

All activations, weights and the per-iteration pre-staged frames stay in HBM; nothing crosses
PCIe inside the refinement loop.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ..config import ROT_COORD_CODE, default_config
from ..runtime import Context, DeviceArray, lib

# name, Cout, kernel, stride, pad  (deepIM_flownet.py:63-107)
ENCODER = [
    ("flow_conv1", 64, 7, 2, 3),
    ("conv2", 128, 5, 2, 2),
    ("conv3", 256, 5, 2, 2),
    ("conv3_1", 256, 3, 1, 1),
    ("conv4", 512, 3, 2, 1),
    ("conv4_1", 512, 3, 1, 1),
    ("conv5", 512, 3, 2, 1),
    ("conv5_1", 512, 3, 1, 1),
    ("conv6", 1024, 3, 2, 1),
    ("conv6_1", 1024, 3, 1, 1),
]
SLOPE = 0.1


def _out_hw(h, w, k, s, p):
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


class _Activations(dict):
    """The activation buffers of a bound network. `act["net_input"]` is always the (B,C,H,W) tensor of the reference graph:
    when the zoom front end wrote the channel-blocked form for conv1 (8-channel input, NC8 encoder) it is converted into the
    NCHW buffer on demand — tests and the bench's parity taps read it, the hot path does not."""

    def __init__(self, net):
        dict.__init__(self)
        self._net = net

    def __getitem__(self, key):
        if key == "net_input" and getattr(self._net, "_input_live_h16", False):
            # fp16 mode: the front end wrote fp16 pixel records; their float values (exactly what conv1 multiplies) as NCHW
            n, h = self._net, self._net.ctx.handle
            out = dict.__getitem__(self, "net_input")
            if n.cin == 8:
                lib.deepim_nhwc_f16_to_nchw_f32(h, out, dict.__getitem__(self, "net_input_h8"), n.B, 8, n.H, n.W)
            else:
                if "net_input_tmp" not in self:
                    self["net_input_tmp"] = n.ctx.empty((n.B, 8, n.H, n.W))
                tmp = dict.__getitem__(self, "net_input_tmp")
                lib.deepim_nhwc_f16_to_nchw_f32(h, tmp, dict.__getitem__(self, "net_input_h8"), n.B, 8, n.H, n.W)
                lib.deepim_copy_channels(h, out, 10, 0, tmp, 8, n.B, n.H * n.W)
                lib.deepim_nhwc_f16_to_nchw_f32(h, tmp, dict.__getitem__(self, "net_input_x2"), n.B, 2, n.H, n.W)
                lib.deepim_copy_channels(h, out, 10, 8, tmp, 2, n.B, n.H * n.W)
        if key == "net_input" and getattr(self._net, "_input_live_nc8", False):
            out, src = dict.__getitem__(self, "net_input"), dict.__getitem__(self, "net_input_nc8")
            lib.deepim_relayout_nc8(self._net.ctx.handle, out, src, self._net.B, 8, self._net.H * self._net.W, 0)
        return dict.__getitem__(self, key)


class _Grads(dict):
    """name → gradient in the parameter's own layout. The conv / deconv weight gradients that the LDS-staged kernel leaves
    tap-major (`tm`: name → (raw (Cout,kh*kw,Cin) buffer, Cout, Cin, kh*kw), deepim_conv2d_wgrad_tm) are permuted into the stored
    natural buffer when somebody asks for them; the SGD kernel reads the raw buffers in place, so a training step never does."""

    def __init__(self, ctx, *a, **kw):
        super().__init__(*a, **kw)
        self._ctx, self.tm = ctx, {}

    def __getitem__(self, name):
        out = dict.__getitem__(self, name)
        if name in self.tm:
            raw, cout, cin, khw = self.tm[name]
            lib.deepim_weight_grad_to_natural(self._ctx.handle, out, raw, cout, cin, khw)
        return out

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]


class deepIM_flownet(object):
    def __init__(self):
        self.cfg = None
        self.ctx = None
        self.B = 0
        self.params = {}    # name -> DeviceArray (raw MXNet-layout parameters)
        self.packed = {}    # conv/deconv name -> packed weights
        self.act = _Activations(self)

    # ------------------------------------------------------------------ graph description
    def get_symbol(self, cfg=None, is_train=False):
        if cfg is None:
            cfg = default_config()
        if cfg.network.REGRESSOR_NUM != 1:
            raise Exception("NOT IMPLEMENTED")  # deepIM_flownet.py:748
        self.cfg = cfg
        if is_train:
            return self.get_train_symbol(cfg)
        return self.get_test_symbol_share(cfg)

    def get_train_symbol(self, cfg):
        """Training graph (deepIM_flownet.py:367-546 with get_convs :32-169 and get_loss :170-365): zoom (region from
        mask_gt_observed) → encoder → fc6/fc7 → rot → L2Normalization, trans → inverse ZoomTrans → Transform3D →
        point-matching loss; with network.PRED_FLOW / PRED_MASK also the FlowNetS refinement decoder, Convolution3 →
        upsampling → flow loss against the ZoomFlow-ed labels, and mask_conv3 → mask_upsampling →
        LogisticRegressionOutput against the zoomed mask_gt_observed. forward_train / backward / update run all of it
        resident on the device."""
        n = cfg.network
        t = cfg.train_iter
        if not (t.SE3_PM_LOSS or t.get("SE3_DIST_LOSS", False)):
            raise NotImplementedError("training graph needs train_iter.SE3_PM_LOSS or train_iter.SE3_DIST_LOSS")
        if str(n.ROT_TYPE).upper() != "QUAT":
            # the reference's get_loss builds a 4-output rot head + L2Normalization whatever ROT_TYPE says (deepIM_flownet.py:211-217):
            # an EULER training graph does not exist there either — refuse instead of training the wrong head
            raise NotImplementedError("training graph: network.ROT_TYPE must be 'QUAT' (deepIM_flownet.py:211-217 is quaternion-only)")
        # rot / trans distance losses (deepIM_flownet.py:238-262)
        self.se3_dist_loss = bool(t.get("SE3_DIST_LOSS", False))
        self.se3_pm_loss = bool(t.SE3_PM_LOSS)
        self.trans_loss_type = str(t.get("TRANS_LOSS_TYPE", "L2"))
        if self.se3_dist_loss and self.trans_loss_type not in ("L2", "smooth_L1", "L1"):
            raise Exception("Does not support small_cfg.TRANS_LOSS_TYPE: {}".format(self.trans_loss_type))      # :258-259
        if getattr(n, "FP16_CONV", False) or getattr(n, "X3_CONV", False):
            # encoder_fp16 / encoder_x3 fill only the fp16 / split16 activations; backward() reads the NCHW fp32 ones
            raise NotImplementedError("training graph runs the fp32 convolutions only (network.FP16_CONV / X3_CONV unset)")
        self.get_test_symbol_share(cfg)
        self.is_train = True
        self.nc8 = False          # NCHW activations: what the backward kernels read
        self.two_streams = False  # backward(): weight gradients on a second stream next to the data gradients (False: one stream)
        self.with_mask_head, self.with_flow_head = bool(n.PRED_MASK), bool(n.PRED_FLOW)    # :183, :314
        self.with_decoder = self.with_mask_head or self.with_flow_head
        return self

    def get_test_symbol_share(self, cfg):
        n = cfg.network
        self.input_mask = bool(n.INPUT_MASK)
        self.input_depth = bool(n.INPUT_DEPTH)
        # deepIM_flownet.py:624 / :676 — which heads stay in the test graph
        self.with_mask_head = bool(n.PRED_MASK) and (cfg.TEST.UPDATE_MASK not in ["init", "box_rendered"]
                                                     or not cfg.TEST.FAST_TEST)
        self.with_flow_head = bool(n.PRED_FLOW) and not cfg.TEST.FAST_TEST
        self.with_decoder = self.with_mask_head or self.with_flow_head
        self.cin = 6 + (2 if self.input_depth else 0) + (2 if self.input_mask else 0)
        # BASELINE config 5: conv stack on the fp16 matrix cores (NHWC fp16 activations, fp32 accumulate)
        self.fp16_conv = bool(n.get("FP16_CONV", False))
        # split-fp16 conv path: fp32-grade products (hi·hi + hi·lo + lo·hi, fp32 accumulation) on the fp16 matrix cores for
        # conv2 … conv6_1; conv1 stays on the fp32 kernel. Within the 1e-4 bar (observed ~1e-6), not bit-exact.
        self.x3_conv = bool(n.get("X3_CONV", False)) and not self.fp16_conv
        self.nc8 = bool(cfg.network.get('NC8_CONV', True)) if hasattr(cfg.network, 'get') else True
        # the 3x3 stride-1 encoder layers as fp32 Winograd F(2x2,3x3) (csrc/wino.hip) wherever the layer fills the chip: same fp32
        # arithmetic, 2.25x fewer multiplies, a different summation (<= 1e-5 of the layer's range from the direct sum). Only on
        # the channel-blocked path; network.WINOGRAD_CONV = False keeps the direct kernels.
        self.winograd = bool(cfg.network.get('WINOGRAD_CONV', True)) if hasattr(cfg.network, 'get') else True
        self.H, self.W = cfg.SCALES[0]
        self.K = np.ascontiguousarray(cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
        # Prop-side channel reversal of the means (zoom_image_with_factor.py:79-81)
        self.pixel_means = np.ascontiguousarray(np.asarray(n.PIXEL_MEANS, np.float32).reshape(3)[::-1])
        self.T_means = np.ascontiguousarray(cfg.dataset.trans_means, dtype=np.float32).reshape(3)
        self.T_stds = np.ascontiguousarray(cfg.dataset.trans_stds, dtype=np.float32).reshape(3)
        self.rot_coord = ROT_COORD_CODE[n.ROT_COORD.lower()]
        self.normalize_flow = float(cfg.dataset.NORMALIZE_FLOW)
        # deepIM_flownet.py:715: the rot head has 3 outputs for EULER, 4 otherwise; tester.py:391-398 hands rot_type to RT_transform
        self.rot_type = str(n.ROT_TYPE).upper()
        if self.rot_type not in ("QUAT", "EULER"):
            raise Exception("Unknown rot_type: {}".format(n.ROT_TYPE))          # RT_transform.py:142-143
        self.rot_param = 3 if self.rot_type == "EULER" else 4
        if self.rot_type == "EULER" and (self.fp16_conv or self.x3_conv):
            raise NotImplementedError("network.ROT_TYPE = 'EULER' runs on the fp32 convolution paths only")
        return self

    def arg_shape_dict(self):
        """MXNet parameter names → shapes (what `sym.infer_shape` reports for the weights)."""
        d = {}
        cin = self.cin
        for name, cout, k, s, p in ENCODER:
            d[name + "_weight"] = (cout, cin, k, k)
            d[name + "_bias"] = (cout,)
            cin = cout
        d["fc6_weight"], d["fc6_bias"] = (256, 1024 * 8 * 10), (256,)
        d["fc7_weight"], d["fc7_bias"] = (256, 256), (256,)
        rp = getattr(self, "rot_param", 4)                       # deepIM_flownet.py:715
        d["rot_weight"], d["rot_bias"] = (rp, 256), (rp,)
        d["trans_weight"], d["trans_bias"] = (3, 256), (3,)
        if self.with_decoder:
            d["Convolution1_weight"], d["Convolution1_bias"] = (2, 1024, 3, 3), (2,)
            d["deconv5_weight"], d["deconv5_bias"] = (1024, 512, 4, 4), (512,)
            d["upsample_flow6to5_weight"], d["upsample_flow6to5_bias"] = (2, 2, 4, 4), (2,)
            d["Convolution2_weight"], d["Convolution2_bias"] = (2, 1026, 3, 3), (2,)
            d["deconv4_weight"], d["deconv4_bias"] = (1026, 256, 4, 4), (256,)
            d["upsample_flow5to4_weight"], d["upsample_flow5to4_bias"] = (2, 2, 4, 4), (2,)
        if self.with_mask_head:
            d["mask_conv3_weight"], d["mask_conv3_bias"] = (1, 770, 3, 3), (1,)
            d["mask_upsampling_weight"] = (1, 1, 32, 32)
        if self.with_flow_head:
            d["Convolution3_weight"], d["Convolution3_bias"] = (2, 770, 3, 3), (2,)
            d["upsampling_weight"] = (2, 1, 32, 32)
        return d

    @staticmethod
    def _init_bilinear(shape):
        """MXNet Initializer._init_bilinear (used at deepIM_flownet.py:808-822)."""
        w = np.zeros(int(np.prod(shape)), dtype=np.float32)
        f = np.ceil(shape[3] / 2.0)
        c = (2 * f - 1 - f % 2) / (2.0 * f)
        for i in range(w.size):
            x = i % shape[3]
            y = (i // shape[3]) % shape[2]
            w[i] = (1 - abs(x / f - c)) * (1 - abs(y / f - c))
        return w.reshape(shape)

    def init_weights(self, cfg=None, arg_params=None, aux_params=None, seed=2333, names=None):
        """Seeded synthetic parameters (no checkpoints offline): He-normal conv/FC so activations stay
        O(1) through the 10 layers, rot/trans as the reference initialises them
        (deepIM_flownet.py:795-803), bilinear upsampling kernels (:808-822)."""
        rng = np.random.default_rng(seed)
        arg_params = {} if arg_params is None else arg_params
        shapes = self.arg_shape_dict()
        self.adapt_checkpoint(arg_params, shapes)
        for name, shape in shapes.items():
            if name in arg_params or (names is not None and name not in names):   # `names`: only these (tests)
                continue
            if name.endswith("upsampling_weight"):
                arg_params[name] = self._init_bilinear(shape)
            elif name.endswith("_bias"):
                arg_params[name] = (0.01 * rng.standard_normal(shape)).astype(np.float32)
            elif name == "rot_weight" and getattr(self, "rot_type", "QUAT") == "EULER":
                arg_params[name] = np.zeros(shape, np.float32)      # deepIM_flownet.py:791-792
            elif name == "rot_weight":
                w = rng.random(shape) * 0.01
                w[0, :] = rng.random(shape[1]) + 0.01
                arg_params[name] = w.astype(np.float32)
            elif name == "trans_weight":
                arg_params[name] = (0.01 * rng.standard_normal(shape)).astype(np.float32)
            else:
                fan_in = int(np.prod(shape[1:]))
                if name.startswith("deconv") or name.startswith("upsample_flow"):
                    fan_in = shape[0] * 4  # 2x2 taps reach each output of a k4 s2 transposed conv
                gain = 2.0 / (1 + SLOPE ** 2)
                arg_params[name] = (rng.standard_normal(shape) * np.sqrt(gain / fan_in)).astype(np.float32)
        return arg_params

    @staticmethod
    def adapt_checkpoint(arg_params, shapes):
        """An RGB-pair (6-channel FlowNet) checkpoint under a graph with depth / mask inputs: the extra input channels of
        `flow_conv1_weight` start from zero weights (deepIM_flownet.py:759-773). In place; returns arg_params."""
        w1 = arg_params.get("flow_conv1_weight")
        if w1 is not None and w1.shape[1] < shapes["flow_conv1_weight"][1]:
            w1 = np.asarray(w1.asnumpy() if hasattr(w1, "asnumpy") else w1, dtype=np.float32)
            extra = shapes["flow_conv1_weight"][1] - w1.shape[1]
            arg_params["flow_conv1_weight"] = np.concatenate(
                [w1, np.zeros((w1.shape[0], extra) + w1.shape[2:], np.float32)], axis=1)
        return arg_params

    # ------------------------------------------------------------------ bind
    def bind(self, ctx: Context, batch_size: int, arg_params: dict):
        assert self.cfg is not None, "call get_symbol first"
        self.ctx, self.B = ctx, int(batch_size)
        B, H, W, h = self.B, self.H, self.W, ctx.handle
        shapes = self.arg_shape_dict()
        for name, shape in shapes.items():
            assert name in arg_params, name
            a = np.ascontiguousarray(arg_params[name], dtype=np.float32)
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            self.params[name] = ctx.array(a)
        # one-time weight re-layout into MFMA tile order
        for name, shape in shapes.items():
            if not name.endswith("_weight") or len(shape) != 4 or name.endswith("upsampling_weight"):
                continue
            base = name[: -len("_weight")]
            if base.startswith("deconv") or base.startswith("upsample_flow"):
                cin, cout = shape[0], shape[1]
                nb = lib.load().deepim_deconv_packed_size(cin, cout)
                pk = DeviceArray(ctx, (nb // 4,))
                lib.deepim_deconv_pack_weights(h, pk, self.params[name], cin, cout)
            else:
                cout, cin, kh, kw = shape
                nb = lib.load().deepim_conv_packed_size(cout, cin, kh, kw)
                pk = DeviceArray(ctx, (nb // 4,))
                lib.deepim_conv_pack_weights(h, pk, self.params[name], cout, cin, kh, kw)
            self.packed[base] = pk
        self.packed_wino, self.wino_s2d = {}, set()
        if self.nc8 and getattr(self, "winograd", False) and not (self.fp16_conv or self.x3_conv or getattr(self, "is_train", False)):
            hh, ww, cin = H, W, self.cin
            L = lib.load()
            for li, (name, cout, k, s_, p_) in enumerate(ENCODER):
                if (k, s_, p_) == (3, 1, 1) and L.deepim_conv_wino_preferred(h, B, cin, hh, ww, cout):
                    pk = DeviceArray(ctx, (L.deepim_conv_wino_packed_size(cout, cin) // 4,))
                    lib.deepim_conv_wino_pack_weights(h, pk, self.params[name + "_weight"], cout, cin)
                    self.packed_wino[name] = pk
                elif (k, s_, p_) == (5, 2, 2) and li > 0 and L.deepim_conv_wino_preferred_s2d(h, B, cin, hh, ww, cout):
                    # stride 2 = stride 1 over the four input phases: the layer before writes its output space-to-depth
                    pk = DeviceArray(ctx, (L.deepim_conv_wino_packed_size(cout, 4 * cin) // 4,))
                    lib.deepim_conv_wino_pack_weights_s2d(h, pk, self.params[name + "_weight"], cout, cin)
                    self.packed_wino[name] = pk
                    self.wino_s2d.add(name)
                hh, ww = _out_hw(hh, ww, k, s_, p_)
                cin = cout
        # fc6: the 84 MB weight in MFMA operand order, so the layer is one pass over the weights on the matrix cores
        nb = lib.load().deepim_fc_packed_size(256, 1024 * 8 * 10)
        self.packed["fc6"] = DeviceArray(ctx, (nb // 4,))
        lib.deepim_fc_pack_weights(h, self.packed["fc6"], self.params["fc6_weight"], 256, 1024 * 8 * 10)
        if self.fp16_conv:   # fp16 weights in MFMA octet order, one-time
            self.packed_f16 = {}
            cin = self.cin
            for name, cout, k, s_, p_ in ENCODER:
                cpad = (cin + 7) // 8 * 8
                nb = lib.load().deepim_conv_f16_packed_size(cout, cpad, k, k)
                pk = DeviceArray(ctx, (nb // 2,), dtype=np.float16)
                lib.deepim_conv_f16_pack_weights(h, pk, self.params[name + "_weight"], cout, cin, cpad, k, k)
                self.packed_f16[name] = pk
                cin = cout
            if self.cin == 8 and self.W % 4 == 0:   # conv1 on the patch kernel, straight from the NCHW fp32 net input
                pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
                lib.deepim_conv1_x3_pack_weights(h, pk, self.params[ENCODER[0][0] + "_weight"], ctypes.c_float(1.0))
                self.packed_f16["conv1_patch"] = pk
            elif self.cin == 10 and self.W % 4 == 0:   # RGB-D input (config 5): the 8 + 2 channel form of the same kernel
                pk = DeviceArray(ctx, (lib.load().deepim_conv1_f16_c10_packed_size() // 2,), dtype=np.float16)
                lib.deepim_conv1_f16_c10_pack_weights(h, pk, self.params[ENCODER[0][0] + "_weight"])
                self.packed_f16["conv1_patch"] = pk
        if self.x3_conv:     # split-fp16 weights [hi 16 | lo 16] in MFMA octet order, scaled by a power of two into fp16's range
            self.packed_x3, self.x3_wscale = {}, {}
            if self.cin == 8 and self.W % 4 == 0:    # conv1 on the split-fp16 patch kernel (8-channel input; otherwise fp32 conv1)
                name = ENCODER[0][0]
                m = float(np.abs(np.asarray(arg_params[name + "_weight"], np.float32)).max())
                self.x3_wscale[name] = 2.0 ** int(np.floor(np.log2(1536.0 / m))) if m > 0 else 1.0
                pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
                lib.deepim_conv1_x3_pack_weights(h, pk, self.params[name + "_weight"], ctypes.c_float(self.x3_wscale[name]))
                self.packed_x3[name] = pk
            cin = ENCODER[0][1]
            for name, cout, k, s_, p_ in ENCODER[1:]:
                m = float(np.abs(np.asarray(arg_params[name + "_weight"], np.float32)).max())
                self.x3_wscale[name] = 2.0 ** int(np.floor(np.log2(1536.0 / m))) if m > 0 else 1.0
                nb = lib.load().deepim_conv_x3_packed_size(cout, cin, k, k)
                pk = DeviceArray(ctx, (nb // 2,), dtype=np.float16)
                lib.deepim_conv_x3_pack_weights(h, pk, self.params[name + "_weight"], cout, cin, k, k,
                                                ctypes.c_float(self.x3_wscale[name]))
                self.packed_x3[name] = pk
                cin = cout
        # activations
        A = self.act
        A["net_input"] = ctx.empty((B, self.cin, H, W))
        self._input_live_nc8 = False
        A["zoom_factor"] = ctx.empty((B, 4))
        hh, ww, cin = H, W, self.cin
        self.enc_geom = []
        for name, cout, k, s, p in ENCODER:
            ho, wo = _out_hw(hh, ww, k, s, p)
            A[name] = ctx.empty((B, cout, ho, wo))
            self.enc_geom.append((name, cin, hh, ww, cout, k, s, p))
            if self.fp16_conv:
                A[name + "_h"] = ctx.empty((B, ho, wo, cout), dtype=np.float16)   # NHWC fp16
            if self.x3_conv:
                A[name + "_x"] = ctx.empty((B, ho, wo, 2 * cout), dtype=np.float16)   # split16 NHWC
            hh, ww, cin = ho, wo, cout
        if self.fp16_conv:
            self.cin_pad = (self.cin + 7) // 8 * 8
            A["net_input_h"] = ctx.empty((B, H, W, self.cin_pad), dtype=np.float16)
            if "conv1_patch" in self.packed_f16 and self.input_mask:
                # the zoom front end writes conv1's fp16 pixel records itself (no fp32 net input, no conversion pass);
                # `net.fp16_fused_input = False` restores the two-step form for A/B measurements
                A["net_input_h8"] = ctx.empty((B, H, W, 8), dtype=np.float16)
                if self.cin == 10:
                    A["net_input_x2"] = ctx.empty((B, H, W, 2), dtype=np.float16)
                self.fp16_fused_input = True
        self._input_live_h16 = False
        A["fc6"], A["fc7"], A["se3"] = ctx.empty((B, 256)), ctx.empty((B, 256)), ctx.empty((B, getattr(self, "rot_param", 4) + 3))
        if getattr(self, "rot_type", "QUAT") == "EULER":
            A["rot_e"], A["zoom_trans_e"], A["trans_e"] = ctx.empty((B, 3)), ctx.empty((B, 3)), ctx.empty((B, 3))
        A["pose_est"] = ctx.empty((B, 3, 4))
        if self.with_decoder:
            A["flow6"] = ctx.empty((B, 2, 8, 10))
            A["Concat2"] = ctx.empty((B, 1026, 15, 20))
            A["flow5"] = ctx.empty((B, 2, 15, 20))
            A["Concat3"] = ctx.empty((B, 770, 30, 40))
        if self.with_mask_head:
            A["mask_lowres"] = ctx.empty((B, 1, 30, 40))
            A["mask_logits"] = ctx.empty((B, 1, H, W))
            A["mask_observed_pred"] = ctx.empty((B, 1, H, W))
        if self.with_flow_head:
            A["flow_lowres"] = ctx.empty((B, 2, 30, 40))
            A["zoom_flow_est"] = ctx.empty((B, 2, H, W))
            A["flow_est"] = ctx.empty((B, 2, H, W))
        ctx.sync()
        return self

    # ------------------------------------------------------------------ forward pieces
    def _conv(self, name, src, dst, B, cin, h, w, cout, k, s, p, slope, ctotal=0, coff=0):
        lib.deepim_conv2d_forward(self.ctx.handle, dst, src, self.packed[name], self.params[name + "_bias"], B, cin, h,
                                  w, cout, k, k, s, p, ctypes.c_float(slope), ctotal, coff)

    def _deconv(self, name, src, dst, B, cin, h, w, cout, ho, wo, slope, ctotal, coff):
        lib.deepim_deconv4x4s2_crop_forward(self.ctx.handle, dst, src, self.packed[name], self.params[name + "_bias"],
                                            B, cin, h, w, cout, ho, wo, 1, 1, ctypes.c_float(slope), ctotal, coff)

    def _conv1_from_nc8(self):
        """OPT-IN (`net.conv1_nc8 = True` after bind; default off): conv1 reads a channel-blocked net input written by the
        zoom front end — the shipped 8-channel graph on the NC8 fp32 encoder (not the fp16 / x3 modes, which convert the NCHW
        input themselves, and not the training graph, whose backward reads NCHW). Measured in round 3 (profiles/r03_conv1_nc8.md):
        conv1 1.016 ms vs 1.006 ms on the NCHW register-fed kernel and the front end 0.179 vs 0.149 ms — conv1 is bound by its
        short K loop (49 tap groups per tile), not by the gather shape, so the default stays NCHW."""
        return (getattr(self, "conv1_nc8", False) and self.nc8 and self.cin == 8 and self.input_mask and not self.input_depth
                and not self.fp16_conv and not self.x3_conv and not getattr(self, "is_train", False) and self.W % 4 == 0)

    def zoom(self, data):
        A, h = self.act, self.ctx.handle
        if self.fp16_conv and getattr(self, "fp16_fused_input", False) and "net_input_h8" in A:
            x2 = dict.__getitem__(A, "net_input_x2") if self.cin == 10 else None
            lib.deepim_zoom_concat_forward_h16(h, data["image_observed"], data["image_rendered"], data["mask_observed"],
                                               data["mask_rendered"], data.get("depth_observed") if self.input_depth else None,
                                               data.get("depth_rendered") if self.input_depth else None, data["src_pose"],
                                               self.K, self.pixel_means, dict.__getitem__(A, "net_input_h8"), x2,
                                               A["zoom_factor"], self.B, self.H, self.W)
            self._input_live_h16, self._input_live_nc8 = True, False
            return
        self._input_live_h16 = False
        if self._conv1_from_nc8():
            if "net_input_nc8" not in A:       # opt-in path: allocated on first use (run once eagerly before a graph capture)
                A["net_input_nc8"] = self.ctx.empty((self.B, self.H, self.W, 8))
            lib.deepim_zoom_concat_forward_nc8(h, data["image_observed"], data["image_rendered"], data["mask_observed"],
                                               data["mask_rendered"], data["src_pose"], self.K, self.pixel_means,
                                               dict.__getitem__(A, "net_input_nc8"), A["zoom_factor"], self.B, self.H, self.W)
            self._input_live_nc8 = True
            return
        self._input_live_nc8 = False
        lib.deepim_zoom_concat_forward(
            h, data["image_observed"], data["image_rendered"],
            data["mask_observed"] if self.input_mask else None, data["mask_rendered"] if self.input_mask else None,
            data.get("depth_observed") if self.input_depth else None,
            data.get("depth_rendered") if self.input_depth else None,
            data["src_pose"], self.K, self.pixel_means, A["net_input"], A["zoom_factor"], self.B, self.H, self.W)

    def encoder_fp16(self):
        """Same 10 layers on the fp16 matrix cores: NCHW fp32 net input → NHWC fp16 → convs → conv6_1 back to
        NCHW fp32 for the (fp32) FC head."""
        A, h, B = self.act, self.ctx.handle, self.B
        geom = self.enc_geom
        if "conv1_patch" in self.packed_f16:
            name = geom[0][0]
            if getattr(self, "_input_live_h16", False):
                lib.deepim_conv1_f16_h16_forward(h, A[name + "_h"], dict.__getitem__(A, "net_input_h8"),
                                                 dict.__getitem__(A, "net_input_x2") if self.cin == 10 else None,
                                                 self.packed_f16["conv1_patch"], self.params[name + "_bias"], B, self.H, self.W,
                                                 ctypes.c_float(SLOPE))
            else:
                conv1 = lib.deepim_conv1_f16_c10_forward if self.cin == 10 else lib.deepim_conv1_f16_forward
                conv1(h, A[name + "_h"], A["net_input"], self.packed_f16["conv1_patch"], self.params[name + "_bias"], B, self.H,
                      self.W, ctypes.c_float(SLOPE))
            src, geom = A[name + "_h"], geom[1:]
        else:
            lib.deepim_nchw_f32_to_nhwc_f16(h, A["net_input_h"], A["net_input"], B, self.cin, self.H, self.W, self.cin_pad)
            src = A["net_input_h"]
        for name, cin, hh, ww, cout, k, s, p in geom:
            cpad = (cin + 7) // 8 * 8
            lib.deepim_conv2d_f16_forward(h, A[name + "_h"], src, self.packed_f16[name], self.params[name + "_bias"], B,
                                          cpad, hh, ww, cout, k, k, s, p, ctypes.c_float(SLOPE))
            src = A[name + "_h"]
        lib.deepim_nhwc_f16_to_nchw_f32(h, A["conv6_1"], src, B, 1024, 8, 10)

    X3_ACT_SCALE = 16.0   # stored split16 activations = value · 16: full 22-bit pairs down to |v| = 2^-6, saturation at 3750

    def encoder_x3(self):
        """The 10 layers with conv2 … conv6_1 on the split-fp16 kernel: conv1 in fp32 (NCHW) → split16 NHWC → 9 x3 convs →
        conv6_1 back to NCHW fp32 for the FC head."""
        A, h, B = self.act, self.ctx.handle, self.B
        c, sa = ctypes.c_float, self.X3_ACT_SCALE
        name, cin, hh, ww, cout, k, s, p = self.enc_geom[0]
        if name in self.packed_x3:
            lib.deepim_conv1_x3_forward(h, A[name + "_x"], A["net_input"], self.packed_x3[name], self.params[name + "_bias"], B,
                                        hh, ww, c(SLOPE), c(sa), c(1.0 / (sa * self.x3_wscale[name])), c(sa))
        else:
            lib.deepim_conv2d_forward_split16(h, A[name + "_x"], A["net_input"], self.packed[name], self.params[name + "_bias"],
                                              B, cin, hh, ww, cout, k, k, s, p, c(SLOPE), c(sa))
        src = A[name + "_x"]
        for name, cin, hh, ww, cout, k, s, p in self.enc_geom[1:]:
            lib.deepim_conv2d_x3_forward(h, A[name + "_x"], src, self.packed_x3[name], self.params[name + "_bias"], B, cin, hh,
                                         ww, cout, k, k, s, p, c(SLOPE), c(1.0 / (sa * self.x3_wscale[name])), c(sa))
            src = A[name + "_x"]
        lib.deepim_split16_to_nchw_f32(h, A["conv6_1"], src, B, 1024, 8, 10, c(1.0 / sa))
        self.act_layout = "x3"

    def encoder(self):
        """10 conv layers. With `self.nc8` (default) the layers hand each other channel-blocked activations
        ([n][C/8][h][w][8]): conv1 reads the NCHW net input and writes NC8, conv6_1 returns to NCHW for fc6 (MXNet's
        flatten order); `self.act[name]` of the layers in between then holds NC8 data (`activation_nchw` converts).
        `self.nc8 = False` keeps NCHW throughout (the canonical-order, bit-exact configuration of the tests)."""
        if self.fp16_conv:
            return self.encoder_fp16()
        if self.x3_conv:
            return self.encoder_x3()
        A = self.act
        src = self.conv1_input()
        for li in range(len(self.enc_geom)):
            self.encoder_layer(li, src)
            src = A[self.enc_geom[li][0]]
        self.act_layout = "nc8" if self.nc8 else "nchw"

    def conv1_input(self):
        """What conv1 reads: the channel-blocked (B,H,W,8) records when the last zoom() wrote them, else the NCHW net input."""
        if getattr(self, "_input_live_nc8", False):
            return dict.__getitem__(self.act, "net_input_nc8")
        return dict.__getitem__(self.act, "net_input")

    def encoder_layer(self, li, src):
        """One encoder conv (index into enc_geom) from `src` into its activation buffer, in the configured layout."""
        name, cin, h, w, cout, k, s, p = self.enc_geom[li]
        out_mode = self._enc_out_mode(li)
        if self.nc8 and name in self.packed_wino:
            fwd = lib.deepim_conv2d_wino_forward_s2d if name in self.wino_s2d else lib.deepim_conv2d_wino_forward
            fwd(self.ctx.handle, self.act[name], src, self.packed_wino[name], self.params[name + "_bias"],      # s2d: 5x5 stride 2 over
                self.B, cin, h, w, cout, ctypes.c_float(SLOPE), out_mode, 0, 0)                                  # the space-to-depth tensor
        elif self.nc8:
            in8 = 1 if (li > 0 or src.shape == (self.B, self.H, self.W, 8)) else 0     # conv1: NC8 records from the zoom front end
            lib.deepim_conv2d_forward_ex(self.ctx.handle, self.act[name], src, self.packed[name], self.params[name + "_bias"],
                                         self.B, cin, h, w, cout, k, k, s, p, ctypes.c_float(SLOPE), 0, 0,
                                         in8, out_mode)
        else:
            self._conv(name, src, self.act[name], self.B, cin, h, w, cout, k, s, p, SLOPE)

    def _enc_out_mode(self, li):
        """Output layout of encoder layer li on the channel-blocked path: 0 = NCHW (the last layer, for fc6), 3 = NC8 in
        space-to-depth order (the next layer is a stride-2 Winograd layer), 1 = NC8."""
        if li == len(self.enc_geom) - 1:
            return 0
        nxt = self.enc_geom[li + 1][0]
        return 3 if self.nc8 and nxt in getattr(self, "wino_s2d", ()) and nxt in getattr(self, "packed_wino", {}) else 1

    def activation_nchw(self, name):
        """Encoder activation `name` as an NCHW device array (a converted copy when the encoder ran channel-blocked)."""
        a = self.act[name]
        names = [g[0] for g in self.enc_geom]
        if getattr(self, "act_layout", "nchw") == "x3" and name in names[:-1]:
            lib.deepim_split16_to_nchw_f32(self.ctx.handle, a, self.act[name + "_x"], a.shape[0], a.shape[1], a.shape[2],
                                           a.shape[3], ctypes.c_float(1.0 / self.X3_ACT_SCALE))
            return a
        if self.fp16_conv or getattr(self, "act_layout", "nchw") != "nc8" or name not in names[:-1]:
            return a
        key = name + "_nchw"          # one conversion buffer per tensor, allocated on first use (not during graph capture)
        if key not in self.act:
            self.act[key] = self.ctx.empty(a.shape)
        out = self.act[key]
        if self._enc_out_mode(names.index(name)) == 3:
            lib.deepim_relayout_nc8_s2d(self.ctx.handle, out, a, a.shape[0], a.shape[1], a.shape[2], a.shape[3], 0)
            return out
        lib.deepim_relayout_nc8(self.ctx.handle, out, a, a.shape[0], a.shape[1], a.shape[2] * a.shape[3], 0)
        return out

    FC6_PLAIN_MAX_BATCH = 4   # measured (tools/bench_fc6.py): B = 1 / 4 / 32 → plain 20 / 24 / 79 µs, packed MFMA 31 / 31 / 32 µs

    def _fc6(self, flat):
        """fc6 (81920 → 256): up to FC6_PLAIN_MAX_BATCH pairs the plain kernel streams the raw weights once faster than the MFMA
        kernel runs on the packed ones (and a training step then needs no 38 µs re-pack of 84 MB); larger batches take the packed
        MFMA path. Fixed by the batch size, so every rank and run sums in the same order."""
        A, P, h, B = self.act, self.params, self.ctx.handle, self.B
        if B <= self.FC6_PLAIN_MAX_BATCH:
            lib.deepim_fc_forward(h, A["fc6"], flat, P["fc6_weight"], P["fc6_bias"], B, flat.shape[1], 256, ctypes.c_float(SLOPE))
        else:
            lib.deepim_fc_forward_packed(h, A["fc6"], flat, self.packed["fc6"], P["fc6_bias"], B, flat.shape[1], 256,
                                         ctypes.c_float(SLOPE))

    def pose_head(self):
        A, P, h, B = self.act, self.params, self.ctx.handle, self.B
        flat = A["conv6_1"].reshape((B, -1))
        self._fc6(flat)
        lib.deepim_fc_forward(h, A["fc7"], A["fc6"], P["fc7_weight"], P["fc7_bias"], B, 256, 256,
                              ctypes.c_float(SLOPE))
        if self.rot_type == "EULER":      # 3-output rot head (deepIM_flownet.py:715-726): FullyConnected x2, inverse ZoomTrans, Concat
            c1 = ctypes.c_float(1.0)
            lib.deepim_fc_forward(h, A["rot_e"], A["fc7"], P["rot_weight"], P["rot_bias"], B, 256, 3, c1)
            lib.deepim_fc_forward(h, A["zoom_trans_e"], A["fc7"], P["trans_weight"], P["trans_bias"], B, 256, 3, c1)
            lib.deepim_zoom_trans_forward(h, A["zoom_factor"], A["zoom_trans_e"], A["trans_e"], 1, B)
            lib.deepim_copy_channels(h, A["se3"], 6, 0, A["rot_e"], 3, B, 1)
            lib.deepim_copy_channels(h, A["se3"], 6, 3, A["trans_e"], 3, B, 1)
            return
        lib.deepim_pose_head_forward(h, A["se3"], A["fc7"], P["rot_weight"], P["rot_bias"], P["trans_weight"],
                                     P["trans_bias"], A["zoom_factor"], B, 256)

    def pose_head_update(self, src_pose, pose_out=None):
        """pose_head() + pose_update() with everything behind fc6 in ONE launch (deepim_pose_tail_forward: fc7 → rot / trans +
        inverse ZoomTrans → se3 → RT_transform); bit-identical to the separate calls, fills the same activations (fc6, fc7,
        se3) and returns the refined poses. `pose_out` may be `src_pose` (in-place update, as the refinement loop does)."""
        A, P, h, B = self.act, self.params, self.ctx.handle, self.B
        out = A["pose_est"] if pose_out is None else pose_out
        if self.rot_type == "EULER":      # the fused tail is the quaternion form; EULER takes the separate launches
            self.pose_head()
            return self.pose_update(src_pose, out)
        self._fc6(A["conv6_1"].reshape((B, -1)))
        lib.deepim_pose_tail_forward(h, A["fc7"], A["se3"], out, A["fc6"], P["fc7_weight"], P["fc7_bias"], P["rot_weight"],
                                     P["rot_bias"], P["trans_weight"], P["trans_bias"], A["zoom_factor"], src_pose, self.T_means,
                                     self.T_stds, self.rot_coord, B, 256, ctypes.c_float(SLOPE))
        return out

    def _skip_into(self, cat, ctotal, name, C, hw):
        """Concat's first input: encoder activation `name` into channels [0, C) of the concat tensor — straight from the channel-blocked
        tensor where the encoder ran channel-blocked (one pass instead of relayout + 2-D blit), else a slice copy."""
        A, h, B = self.act, self.ctx.handle, self.B
        names = [g[0] for g in self.enc_geom]
        if (getattr(self, "act_layout", "nchw") == "nc8" and not self.fp16_conv and name in names[:-1]
                and self._enc_out_mode(names.index(name)) != 3):
            lib.deepim_relayout_nc8_slice(h, A[cat], ctotal, 0, A[name], B, C, hw)
        else:
            lib.deepim_copy_channels(h, A[cat], ctotal, 0, self.activation_nchw(name), C, B, hw)

    def decoder(self):
        """FlowNetS refinement (deepIM_flownet.py:120-167)."""
        A, h, B = self.act, self.ctx.handle, self.B
        self._conv("Convolution1", A["conv6_1"], A["flow6"], B, 1024, 8, 10, 2, 3, 1, 1, 1.0)
        self._skip_into("Concat2", 1026, "conv5_1", 512, 15 * 20)
        self._deconv("deconv5", A["conv6_1"], A["Concat2"], B, 1024, 8, 10, 512, 15, 20, SLOPE, 1026, 512)
        self._deconv("upsample_flow6to5", A["flow6"], A["Concat2"], B, 2, 8, 10, 2, 15, 20, 1.0, 1026, 1024)
        self._conv("Convolution2", A["Concat2"], A["flow5"], B, 1026, 15, 20, 2, 3, 1, 1, 1.0)
        self._skip_into("Concat3", 770, "conv4_1", 512, 30 * 40)
        self._deconv("deconv4", A["Concat2"], A["Concat3"], B, 1026, 15, 20, 256, 30, 40, SLOPE, 770, 512)
        self._deconv("upsample_flow5to4", A["flow5"], A["Concat3"], B, 2, 15, 20, 2, 30, 40, 1.0, 770, 768)

    def heads(self):
        A, P, h, B, H, W = self.act, self.params, self.ctx.handle, self.B, self.H, self.W
        # both predictors stream the same 770-channel Concat3: back to back, so that the second finds it where the first left it (the
        # Infinity Cache) instead of behind the 120 MB the first head's upsampling + inverse zoom move (round 6 trace: 62 -> 40 us at B = 32)
        if self.with_mask_head:  # deepIM_flownet.py:627-666
            self._conv("mask_conv3", A["Concat3"], A["mask_lowres"], B, 770, 30, 40, 1, 3, 1, 1, 1.0)
        if self.with_flow_head:  # deepIM_flownet.py:677-713
            self._conv("Convolution3", A["Concat3"], A["flow_lowres"], B, 770, 30, 40, 2, 3, 1, 1, 1.0)
        if self.with_mask_head:
            lib.deepim_upsample16_crop_forward(h, A["mask_logits"], A["mask_lowres"], P["mask_upsampling_weight"], B, 1,
                                               30, 40, H, W, 8, 8, ctypes.c_float(1.0))
            lib.deepim_mask_head_forward(h, A["mask_observed_pred"], None, A["mask_logits"], A["zoom_factor"], B, H, W)
        if self.with_flow_head:
            lib.deepim_upsample16_crop_forward(h, A["zoom_flow_est"], A["flow_lowres"], P["upsampling_weight"], B, 2, 30,
                                               40, H, W, 8, 8, ctypes.c_float(self.normalize_flow))
            lib.deepim_zoom_flow_forward(h, A["zoom_factor"], A["zoom_flow_est"], None, A["flow_est"], None, 1, B, H, W)

    def pose_update(self, src_pose, pose_out=None):
        """RT_transform of every pair (tester.py:391-398)."""
        out = self.act["pose_est"] if pose_out is None else pose_out
        fn = lib.deepim_rt_transform_euler if self.rot_type == "EULER" else lib.deepim_rt_transform   # RT_transform.py:138-143
        fn(self.ctx.handle, out, None, src_pose, self.act["se3"], self.T_means, self.T_stds, self.rot_coord, self.B)
        return out

    def forward(self, data):
        """One network forward = `Predictor.predict` (tester.py:45-47). Returns the output dict (device arrays)."""
        self.zoom(data)
        self.encoder()
        if self.with_decoder:
            self.decoder()
            self.heads()
        self.pose_head()
        out = {"se3": self.act["se3"], "zoom_factor": self.act["zoom_factor"]}
        if self.with_mask_head:
            out["mask_observed_pred"] = self.act["mask_observed_pred"]
        if self.with_flow_head:
            out["flow_est_crop"] = self.act["flow_est"]
        return out

    def capture_iteration(self, data, pose_out=None):
        """Record one refinement iteration (all its launches) into a hipGraph and return the graph id. The
        buffers of `data` / `pose_out` are baked in: refresh their CONTENTS between replays, not the objects.
        Run `refine_iteration` once eagerly first (tap tables, split-K autotuning, scratch growth)."""
        gid = ctypes.c_int(-1)
        lib.deepim_graph_begin(self.ctx.handle)
        try:
            self.refine_iteration(data, pose_out)
        finally:
            lib.deepim_graph_end(self.ctx.handle, ctypes.byref(gid))
        return gid.value

    def replay(self, graph_id):
        lib.deepim_graph_launch(self.ctx.handle, graph_id)

    def refine_iteration(self, data, pose_out=None):
        """One pose-refinement iteration for the bound batch: zoom → network → inverse ZoomTrans →
        RT_transform.  `data["src_pose"]` is the current estimate; returns the refined (B,3,4) poses."""
        self.zoom(data)
        self.encoder()
        if self.with_decoder:
            self.decoder()
            self.heads()
        return self.pose_head_update(data["src_pose"], pose_out)


# ------------------------------------------------------------------------------------------------- training ----
def _train_methods():
    """Training-side methods of deepIM_flownet (kept below the inference code; attached to the class at import)."""

    def bind_train(self, ctx, batch_size, arg_params, num_points=3000):
        """bind() + gradient, momentum and workspace buffers for one training-style iteration."""
        assert getattr(self, "is_train", False), "call get_symbol(cfg, is_train=True) first"
        self.bind(ctx, batch_size, arg_params)
        B, H, W = self.B, self.H, self.W
        # rot / trans FullyConnected run their backward as ONE 7-row layer: their weights and gradients live as row ranges of
        # shared (7,256) / (7,) buffers, so nothing is assembled or split per step
        self.side = Context(ctx.device_id) if self.two_streams else None     # second stream of the backward (see backward())
        w7, dw7, db7 = ctx.empty((7, 256)), ctx.zeros((7, 256)), ctx.zeros((7,))
        w7[0:4].copyfrom(self.params["rot_weight"]); w7[4:7].copyfrom(self.params["trans_weight"])
        self.params["rot_weight"], self.params["trans_weight"] = w7[0:4], w7[4:7]
        self.grad = _Grads(ctx, {name: ctx.zeros(a.shape) for name, a in self.params.items()})
        self._upd_ws = None        # train_step's batch-updater workspace is sized for the bound batch
        self._sgd_table = None     # its rows hold raw pointers of the buffers allocated here: never reuse one across binds
        # tap-major weight gradients (deepim_conv2d_wgrad_tm) only where that entry applies: Cin % 8 == 0 and the LDS-staged
        # kernel selected on this context. Everything else (conv1 of the 6- / 10-channel inputs, wgrad_lds = 0) takes
        # deepim_conv2d_wgrad into the natural buffer and the SGD table reads that one (layout word 0).
        opt = ctypes.c_int(0)
        lib.deepim_get_option(ctx.handle, b"wgrad_lds", ctypes.byref(opt))
        for gname, cin, _h, _w, cout, k, _s, _p in self.enc_geom:       # conv layers: (Cout, Cin, k, k)
            if opt.value and cin % 8 == 0:
                self.grad.tm[gname + "_weight"] = (ctx.zeros((cout, k * k, cin)), cout, cin, k * k)
        if self.with_decoder and opt.value:                              # deconvs: MXNet (cin, cout, 4, 4) = filters cin, channels cout
            for gname, cin, cout in (("deconv4", 1026, 256), ("deconv5", 1024, 512)):
                self.grad.tm[gname + "_weight"] = (ctx.zeros((cin, 16, cout)), cin, cout, 16)
        self.grad["rot_weight"], self.grad["trans_weight"] = dw7[0:4], dw7[4:7]
        self.grad["rot_bias"], self.grad["trans_bias"] = db7[0:4], db7[4:7]
        self.mom = {name: ctx.zeros(a.shape) for name, a in self.params.items()}
        A = self.act
        A["rot"], A["rot_norm"] = ctx.empty((B, 4)), ctx.empty((B, 4))
        A["zoom_trans"], A["trans_est"] = ctx.empty((B, 3)), ctx.empty((B, 3))
        A["points_est"] = ctx.empty((B, 3, num_points))
        A["pm_loss"], A["pm_loss_sum"] = ctx.empty((B, 3, num_points)), ctx.empty((1,))
        self.num_points = num_points
        # backward workspaces: two ping-pong activation-gradient buffers, the un-cropped gradient of the decoder's transposed
        # convolutions, and the packed data-gradient weights (sized for the largest layer)
        big = max(int(np.prod(A[g[0]].shape)) for g in self.enc_geom)
        self.ws = {"ga": ctx.empty((big,)), "gb": ctx.empty((big,))}
        dil, pmax = 4, 4
        psize = lib.load().deepim_conv_packed_size
        for name, cin, h, w, cout, k, s_, p_ in self.enc_geom[1:]:
            pmax = max(pmax, psize(cin, cout, k, k) // 4)
            pmax = max(pmax, lib.load().deepim_conv_dgrad_packed_size(cout, cin, k, s_, p_) // 4)
        if self.with_decoder:
            # the un-cropped gradients of the two big transposed convolutions, and the role-swapped weights of the decoder layers
            dil = max(dil, B * 256 * 32 * 42, B * 512 * 18 * 22)
            pmax = max(pmax, psize(1026, 256, 4, 4) // 4, psize(1024, 512, 4, 4) // 4, psize(1026, 2, 3, 3) // 4,
                       psize(770, 2, 3, 3) // 4)
            W_ = self.ws
            W_["d_Concat3"], W_["d_Concat2"] = ctx.empty((B, 770, 30, 40)), ctx.empty((B, 1026, 15, 20))
            W_["d_skip4"], W_["d_skip5"] = ctx.empty((B, 512, 30, 40)), ctx.empty((B, 512, 15, 20))
            W_["d_dec61"] = ctx.empty((B, 1024, 8, 10))
            W_["d_flow5"], W_["d_flow6"] = ctx.empty((B, 2, 15, 20)), ctx.empty((B, 2, 8, 10))
            W_["d_low"] = ctx.empty((B, 2, 30, 40))
        if self.with_flow_head:
            A["flow_loss"], A["flow_loss_sum"] = ctx.empty((B, 2, H, W)), ctx.empty((1,))
            A["zoom_flow_gt"], A["zoom_flow_weights"] = ctx.empty((B, 2, H, W)), ctx.empty((B, 2, H, W))
            self.ws["d_flow_hi"] = ctx.empty((B, 2, H, W))
        if self.with_mask_head:
            A["mask_prob"], A["zoom_mask_gt_observed"] = ctx.empty((B, 1, H, W)), ctx.empty((B, 1, H, W))
            self.ws["zm_a"], self.ws["zm_b"], self.ws["zm_f"] = ctx.empty((B, 1, H, W)), ctx.empty((B, 1, H, W)), ctx.empty((B, 4))
            self.ws["d_mask_hi"] = ctx.empty((B, 1, H, W))
        self.ws["dil"], self.ws["wt_packed"] = ctx.empty((dil,)), ctx.empty((pmax,))
        self.ws["g256a"], self.ws["g256b"] = ctx.empty((B, 256)), ctx.empty((B, 256))
        self.ws["dy7"], self.ws["w7"], self.ws["dw7"], self.ws["db7"] = ctx.empty((B, 7)), w7, dw7, db7
        self.ws["d_points"] = ctx.empty((B, 3, num_points))
        self.ws["d_rot_norm"], self.ws["d_trans_est"] = ctx.empty((B, 4)), ctx.empty((B, 3))
        self.ws["d_rot"], self.ws["d_trans"] = ctx.empty((B, 4)), ctx.empty((B, 3))
        if self.se3_dist_loss:            # deepIM_flownet.py:238-262
            A["zoom_trans_gt"], A["rot_loss"] = ctx.empty((B, 3)), ctx.empty((B,))
            A["trans_loss"], A["trans_loss_sum"] = ctx.empty((B, 3, 1)), ctx.empty((1,))
            self.ws["d_rot_norm_dist"], self.ws["d_zoom_trans_dist"] = ctx.empty((B, 4)), ctx.empty((B, 3, 1))
        ctx.sync()
        return self

    def forward_train(self, data, label):
        """data: image_observed, image_rendered, mask_observed, mask_rendered [, depth_*], src_pose; label:
        mask_gt_observed, point_cloud_model, point_cloud_weights, point_cloud_observed [, flow, flow_weights] (device arrays).
        Returns the point-matching loss sum (device scalar) after filling every activation the backward needs; the flow loss
        and the mask probability land in self.act["flow_loss"] / ["mask_prob"]."""
        A, P, h, B, H, W = self.act, self.params, self.ctx.handle, self.B, self.H, self.W
        c = ctypes.c_float
        t = self.cfg.train_iter
        lib.deepim_zoom_concat_train_forward(
            h, data["image_observed"], data["image_rendered"], data["mask_observed"] if self.input_mask else None,
            label["mask_gt_observed"] if self.input_mask else None, data["mask_rendered"] if self.input_mask else None,
            data.get("depth_observed") if self.input_depth else None, data.get("depth_rendered") if self.input_depth else None,
            data["src_pose"], self.K, self.pixel_means, A["net_input"], A["zoom_factor"], B, H, W)
        self.encoder()
        if self.with_decoder:
            self.decoder()
        if self.with_flow_head:   # deepIM_flownet.py:183-207 (+ the ZoomFlow of the labels, :478-492)
            self._conv("Convolution3", A["Concat3"], A["flow_lowres"], B, 770, 30, 40, 2, 3, 1, 1, 1.0)
            lib.deepim_upsample16_crop_forward(h, A["zoom_flow_est"], A["flow_lowres"], P["upsampling_weight"], B, 2, 30, 40,
                                               H, W, 8, 8, c(1.0))                      # flow_est_crop, in units of NORMALIZE_FLOW
            lib.deepim_zoom_flow_forward(h, A["zoom_factor"], label["flow"], label["flow_weights"], A["zoom_flow_gt"],
                                         A["zoom_flow_weights"], 0, B, H, W)
            lib.deepim_flow_loss(h, A["flow_loss"], A["flow_loss_sum"], self.ws["d_flow_hi"], A["zoom_flow_est"],
                                 A["zoom_flow_gt"], A["zoom_flow_weights"], c(self.normalize_flow), c(t.LW_FLOW / (480 * 640)),
                                 B * 2 * H * W)
        if self.with_mask_head:   # :314-361 (+ ZoomMask of mask_gt_observed, :395-412)
            self._conv("mask_conv3", A["Concat3"], A["mask_lowres"], B, 770, 30, 40, 1, 3, 1, 1, 1.0)
            lib.deepim_upsample16_crop_forward(h, A["mask_logits"], A["mask_lowres"], P["mask_upsampling_weight"], B, 1, 30, 40,
                                               H, W, 8, 8, c(1.0))
            lib.deepim_zoom_mask_forward(h, data["mask_observed"], label["mask_gt_observed"], data["mask_rendered"],
                                         data["src_pose"], self.K, self.ws["zm_a"], A["zoom_mask_gt_observed"], self.ws["zm_b"],
                                         self.ws["zm_f"], B, H, W)
            # LogisticRegressionOutput: grad = grad_scale / num_output · (p − y), num_output = H·W per sample
            lib.deepim_mask_logistic(h, A["mask_prob"], self.ws["d_mask_hi"], A["mask_logits"], A["zoom_mask_gt_observed"],
                                     c(t.LW_MASK / (H * W)), B * H * W)
        flat = A["conv6_1"].reshape((B, -1))
        self._fc6(flat)
        lib.deepim_fc_forward(h, A["fc7"], A["fc6"], P["fc7_weight"], P["fc7_bias"], B, 256, 256, c(SLOPE))
        lib.deepim_fc_forward(h, A["rot"], A["fc7"], P["rot_weight"], P["rot_bias"], B, 256, 4, c(1.0))
        lib.deepim_fc_forward(h, A["zoom_trans"], A["fc7"], P["trans_weight"], P["trans_bias"], B, 256, 3, c(1.0))
        lib.deepim_l2_normalize_forward(h, A["rot_norm"], A["rot"], B, 4, c(1e-10))                       # :217
        lib.deepim_zoom_trans_forward(h, A["zoom_factor"], A["zoom_trans"], A["trans_est"], 1, B)           # :218-225
        ltypes = {"L1": 0, "L2": 1, "smooth_L1": 2}
        if self.se3_dist_loss:            # :238-262: rot_loss = 1 - (q_gt . q_est)^2 (grad_scale LW_ROT), trans loss on the ZOOMED deltas
            rot_gt = label["rot"] if "rot" in label else label["rot_gt"]            # Variable(name="rot") / ("trans"), :452-453
            trans_gt = label["trans"] if "trans" in label else label["trans_gt"]
            lib.deepim_zoom_trans_forward(h, A["zoom_factor"], trans_gt, A["zoom_trans_gt"], 0, B)         # :455-457
            lib.deepim_rot_dist_loss(h, A["rot_loss"], self.ws["d_rot_norm_dist"], rot_gt, A["rot_norm"], c(t.LW_ROT), B)
            lib.deepim_point_matching_loss(h, A["trans_loss"], A["trans_loss_sum"], self.ws["d_zoom_trans_dist"], A["zoom_trans"],
                                           A["zoom_trans_gt"], None, c(1.0), ltypes[self.trans_loss_type],
                                           c(t.get("TRANS_SMOOTH_L1_SCALAR", 3.0)), c(t.LW_TRANS), B, 1)
        if self.se3_pm_loss:
            lib.deepim_transform3d_forward(h, A["points_est"], label["point_cloud_model"], A["rot_norm"], A["trans_est"],
                                           data["src_pose"], self.T_means, self.T_stds, self.rot_coord, B, self.num_points)
            lib.deepim_point_matching_loss(h, A["pm_loss"], A["pm_loss_sum"], self.ws["d_points"], A["points_est"],
                                           label["point_cloud_observed"], label["point_cloud_weights"],
                                           c(self.cfg.dataset.NORMALIZE_3D_POINT), ltypes[t.SE3_PM_LOSS_TYPE], c(t.SE3_PM_SL1_SCALAR),
                                           c(t.LW_PM / t.NUM_3D_SAMPLE), B, self.num_points)                    # :265-312
        self._train_io = (data, label)
        return A["pm_loss_sum"] if self.se3_pm_loss else A["trans_loss_sum"]

    def _dgrad(self, dx, dz, w_raw, B, cin, hh, ww, cout, k, s_, p_, ho, wo, act_y=None, add=None):
        """dx (B,cin,hh,ww) of a Convolution (cout,cin,k,k; stride s_, pad p_) given dz (B,cout,ho,wo); with act_y (the saved output
        of the layer below, [+ add: the gradient arriving over its skip connection]) dx already is that layer's dz =
        lrelu'(act_y)·(dx + add), applied in the convolution's final stores.
        stride 1: the forward MFMA conv kernel on dz with the transposed, flipped weights, pad k-1-p.
        stride 2: four stride-1 convolutions of the UN-dilated dz, one per output parity class (y % 2, x % 2) with the
        sub-kernel of the taps that class meets (k = 3: 1, 2, 2 and 4 taps; k = 5: 4, 6, 6 and 9), each storing its result
        window on its parity positions of dx — exactly the ideal multiply-adds (round 2 convolved a zero-dilated dz: 4x those);
        the four classes are packed, convolved and reduced by one launch each."""
        lib.deepim_conv2d_dgrad(self.ctx.handle, dx, dz, w_raw, self.ws["wt_packed"], B, cin, hh, ww, cout, k, s_, p_, act_y, add,
                                ctypes.c_float(SLOPE))

    def _small_conv_backward(self, name, src, dz, dx, cin, hh, ww, cout):
        """A 3x3 s1 p1 prediction layer (Convolution1/2/3, mask_conv3): gradients into self.grad, data gradient into dx."""
        h, B = self.ctx.handle, self.B
        lib.deepim_conv2d_wgrad_bias(h, self.grad[name + "_weight"], self.grad[name + "_bias"], src, dz, B, cin, hh, ww, cout, 3, 3,
                                     1, 1)
        self._dgrad(dx, dz, self.params[name + "_weight"], B, cin, hh, ww, cout, 3, 1, 1, hh, ww)

    def _head_conv_backward(self, name, d_low, cout, first):
        """Convolution3 / mask_conv3 on Concat3: the first head writes d_Concat3, the second adds to it."""
        W_ = self.ws
        if first:
            self._small_conv_backward(name, self.act["Concat3"], d_low, W_["d_Concat3"], 770, 30, 40, cout)
        else:
            tmp = W_["ga"]
            self._small_conv_backward(name, self.act["Concat3"], d_low, tmp, 770, 30, 40, cout)
            lib.deepim_axpy(self.ctx.handle, W_["d_Concat3"], tmp, ctypes.c_float(1.0), W_["d_Concat3"].size)

    def _deconv_backward(self, name, x, dcat, ycat, ctotal, coff, dx, cin, hh, ww, cout, ho, wo):
        """Deconvolution k4 s2 + Crop(1,1) [+ LeakyReLU] whose output is channels [coff, coff+cout) of a Concat: dcat = gradient of
        the concat (B,ctotal,ho,wo), ycat = the saved concat (None: no activation). Bias / weight gradients into self.grad, data
        gradient (B,cin,hh,ww) into dx. One walk slices the concat gradient, applies the activation gradient, sums the bias gradient
        and puts the result back into the un-cropped frame; the MXNet weight (cin,cout,4,4) already is the Convolution layout of the
        adjoint (filters = cin, channels = cout), so the data gradient is a stride-2 convolution of that frame and the weight
        gradient a conv wgrad with input and output swapped."""
        h, B = self.ctx.handle, self.B
        hf, wf = 2 * hh + 2, 2 * ww + 2
        lib.deepim_slice_lrelu_bias_scatter(h, self.ws["dil"], self.grad[name + "_bias"], dcat, ycat, B, ctotal, coff, cout, ho, wo,
                                            hf, wf, 1, 1, ctypes.c_float(SLOPE))
        if name + "_weight" in self.grad.tm:
            lib.deepim_conv2d_wgrad_tm(h, self.grad.tm[name + "_weight"][0], self.ws["dil"], x, B, cout, hf, wf, cin, 4, 4, 2, 0)
        else:
            lib.deepim_conv2d_wgrad(h, self.grad[name + "_weight"], self.ws["dil"], x, B, cout, hf, wf, cin, 4, 4, 2, 0)
        order = lib.load().deepim_conv_weight_order(h, B, cout, hf, wf, cin, 4, 4, 2, 0)         # packed for this one use
        lib.deepim_conv_pack_weights_ex(h, self.ws["wt_packed"], self.params[name + "_weight"], cin, cout, 4, 4, order)
        lib.deepim_conv2d_forward(h, dx, self.ws["dil"], self.ws["wt_packed"], None, B, cout, hf, wf, cin, 4, 4, 2, 0,
                                  ctypes.c_float(1.0), 0, 0)

    def _decoder_backward(self):
        """Backward of the flow / mask heads and the refinement decoder (deepIM_flownet.py:120-167): leaves the skip
        gradients in d_skip4 / d_skip5 and the conv6_1 gradient in d_dec61."""
        A, P, W_, h, B, H, W = self.act, self.params, self.ws, self.ctx.handle, self.B, self.H, self.W
        c = ctypes.c_float
        ext = lib.deepim_extract_channels
        if self.with_flow_head:   # upsampling (fixed bilinear, lr_mult 0) → Convolution3
            lib.deepim_upsample16_crop_backward(h, W_["d_low"], W_["d_flow_hi"], P["upsampling_weight"], B, 2, 30, 40, H, W, 8, 8,
                                                c(1.0))
            self._head_conv_backward("Convolution3", W_["d_low"], 2, first=True)
        if self.with_mask_head:   # mask_upsampling → mask_conv3
            lib.deepim_upsample16_crop_backward(h, W_["d_low"], W_["d_mask_hi"], P["mask_upsampling_weight"], B, 1, 30, 40, H, W,
                                                8, 8, c(1.0))
            self._head_conv_backward("mask_conv3", W_["d_low"], 1, first=not self.with_flow_head)
        # Concat3 = [conv4_1 | lrelu(deconv4) | upsample_flow5to4]
        ext(h, W_["d_skip4"], W_["d_Concat3"], 770, 0, 512, B, 1200)
        self._deconv_backward("deconv4", A["Concat2"], W_["d_Concat3"], A["Concat3"], 770, 512, W_["d_Concat2"], 1026, 15, 20, 256, 30, 40)
        self._deconv_backward("upsample_flow5to4", A["flow5"], W_["d_Concat3"], None, 770, 768, W_["d_flow5"], 2, 15, 20, 2, 30, 40)
        self._small_conv_backward("Convolution2", A["Concat2"], W_["d_flow5"], W_["ga"], 1026, 15, 20, 2)
        lib.deepim_axpy(h, W_["d_Concat2"], W_["ga"], c(1.0), W_["d_Concat2"].size)
        # Concat2 = [conv5_1 | lrelu(deconv5) | upsample_flow6to5]
        ext(h, W_["d_skip5"], W_["d_Concat2"], 1026, 0, 512, B, 300)
        self._deconv_backward("deconv5", A["conv6_1"], W_["d_Concat2"], A["Concat2"], 1026, 512, W_["d_dec61"], 1024, 8, 10, 512, 15, 20)
        self._deconv_backward("upsample_flow6to5", A["flow6"], W_["d_Concat2"], None, 1026, 1024, W_["d_flow6"], 2, 8, 10, 2, 15, 20)
        self._small_conv_backward("Convolution1", A["conv6_1"], W_["d_flow6"], W_["ga"], 1024, 8, 10, 2)
        lib.deepim_axpy(h, W_["d_dec61"], W_["ga"], c(1.0), W_["d_dec61"].size)

    def backward(self):
        """module.backward (deepim/core/module.py:1131-1137): fills self.grad for every parameter."""
        A, P, G, W_, h, B = self.act, self.params, self.grad, self.ws, self.ctx.handle, self.B
        c = ctypes.c_float
        data, label = self._train_io
        if self.with_decoder:
            self._decoder_backward()
        if self.se3_pm_loss:
            lib.deepim_transform3d_backward(h, W_["d_rot_norm"], W_["d_trans_est"], W_["d_points"], label["point_cloud_model"],
                                            A["rot_norm"], A["trans_est"], data["src_pose"], self.T_means, self.T_stds,
                                            self.rot_coord, B, self.num_points)
            if self.se3_dist_loss:        # both heads of the gradient meet at rot_est_norm (:217 -> :240, :300)
                lib.deepim_axpy(h, W_["d_rot_norm"], W_["d_rot_norm_dist"], c(1.0), B * 4)
        else:
            W_["d_rot_norm"].copyfrom(W_["d_rot_norm_dist"])
        lib.deepim_l2_normalize_backward(h, W_["d_rot"], W_["d_rot_norm"], A["rot"], B, 4, c(1e-10))
        if self.se3_pm_loss:
            lib.deepim_zoom_trans_backward(h, A["zoom_factor"], W_["d_trans_est"], W_["d_trans"], 1, 0, B)     # b_zoom_grad=False
            if self.se3_dist_loss:        # ... and at zoom_trans_est (:212 -> :218, :251)
                lib.deepim_axpy(h, W_["d_trans"], W_["d_zoom_trans_dist"], c(1.0), B * 3)
        else:
            W_["d_trans"].copyfrom(W_["d_zoom_trans_dist"].reshape((B, 3)))
        # rot / trans FullyConnected as one 7-row layer: dy7 = [d_rot | d_trans], w7 = [rot_weight; trans_weight]
        lib.deepim_copy_channels(h, W_["dy7"], 7, 0, W_["d_rot"], 4, B, 1)
        lib.deepim_copy_channels(h, W_["dy7"], 7, 4, W_["d_trans"], 3, B, 1)
        lib.deepim_fc_backward(h, W_["g256a"], W_["dw7"], W_["db7"], W_["dy7"], A["fc7"], W_["w7"], B, 256, 7)   # G[rot_*], G[trans_*] are its rows
        # fc7, fc6 (LeakyReLU gradient from the saved outputs)
        lib.deepim_lrelu_backward(h, W_["g256a"], W_["g256a"], A["fc7"], c(SLOPE), B * 256)
        lib.deepim_fc_backward(h, W_["g256b"], G["fc7_weight"], G["fc7_bias"], W_["g256a"], A["fc6"], P["fc7_weight"], B, 256, 256)
        lib.deepim_lrelu_backward(h, W_["g256b"], W_["g256b"], A["fc6"], c(SLOPE), B * 256)
        n6 = 1024 * 8 * 10
        ga, gb = W_["ga"], W_["gb"]
        lib.deepim_fc_backward(h, ga, G["fc6_weight"], G["fc6_bias"], W_["g256b"], A["conv6_1"].reshape((B, n6)),
                               P["fc6_weight"], B, n6, 256)
        skips = {"conv5_1": "d_skip5", "conv4_1": "d_skip4"} if self.with_decoder else {}
        # encoder, last layer first: dz = lrelu'(y)·(dy [+ the gradient over the layer's skip connection]) and the bias gradient in
        # one fused walk, in place over dy; dx into the other buffer. (_dgrad can apply the activation gradient of the layer below
        # in the convolution's own final stores instead — measured slower, 4.56 → 4.63 ms: the scattered epilogue reads of the
        # saved output cost more than the streaming pass they replace. profiles/r03_train_backward.md)
        # The weight gradient of a layer runs on a SECOND stream (self.side: own context, own scratch) next to the data gradient:
        # both only read dz, and on conv4 … conv6_1 neither fills the chip through its pack → convolution → second-pass chain.
        # Ordering: side waits for dz; main waits for the previous layer's weight gradient before its data gradient overwrites
        # the buffer that one reads (the two gradient buffers ping-pong).
        extra = {"conv6_1": "d_dec61", **skips} if self.with_decoder else {}
        side = self.side.handle if self.side is not None else h
        for li in range(len(self.enc_geom) - 1, -1, -1):
            name, cin_, hh_, ww_, cout_, k_, s_, p_ = self.enc_geom[li]
            ho_, wo_ = _out_hw(hh_, ww_, k_, s_, p_)
            lib.deepim_lrelu_bias_backward(h, ga, G[name + "_bias"], ga, W_[extra[name]] if name in extra else None, A[name],
                                           c(SLOPE), B, cout_, ho_ * wo_)
            lib.deepim_stream_wait(h, side)      # weight gradient of layer li+1 done: gb may be overwritten
            lib.deepim_stream_wait(side, h)      # dz of this layer ready
            src = A["net_input"] if li == 0 else A[self.enc_geom[li - 1][0]]
            if name + "_weight" in G.tm:
                lib.deepim_conv2d_wgrad_tm(side, G.tm[name + "_weight"][0], src, ga, B, cin_, hh_, ww_, cout_, k_, k_, s_, p_)   # tap-major: _Grads
            else:       # Cin % 8 != 0 (6- / 10-channel conv1) or wgrad_lds = 0: natural (Cout, Cin, k, k) gradient
                lib.deepim_conv2d_wgrad(side, dict.__getitem__(G, name + "_weight"), src, ga, B, cin_, hh_, ww_, cout_, k_, k_, s_, p_)
            if li > 0:
                self._dgrad(gb, ga, P[name + "_weight"], B, cin_, hh_, ww_, cout_, k_, s_, p_, ho_, wo_)
            ga, gb = gb, ga
        lib.deepim_stream_wait(h, side)          # every gradient is in place when the main stream goes on (update)
        return G

    def update(self, lr, wd=0.0005, momentum=0.975, rescale_grad=1.0, clip_gradient=None):
        """The "sgd" optimizer step of train.py:296-303 (MXNet sgd_mom_update) on every parameter, then the re-pack of the
        conv / deconv / fc6 weights the forward kernels read. As mx.optimizer does when Module hands it the parameter
        names: weight decay only on `*_weight` (wd_mult 0 elsewhere), and the bilinear upsampling kernels stay fixed
        (attr lr_mult 0, deepIM_flownet.py:193,333,643,695)."""
        h = self.ctx.handle
        c = ctypes.c_float
        tab = getattr(self, "_sgd_table", None)
        if tab is None or tab[0] != float(wd):
            # one launch for all parameters: rows {w, mom, g, n, wd bits | first block << 32, layout of g} (deepim_sgd_mom_update_multi)
            rows, block = [], 0
            for name, w in self.params.items():
                if name.endswith("upsampling_weight"):
                    continue
                wd_n = wd if name.endswith("_weight") else 0.0
                wd_bits = int(np.array([wd_n], np.float32).view(np.uint32)[0])
                if name in self.grad.tm:       # tap-major gradient, read in place: layout word = Cin | kh*kw << 32
                    raw, _co, cin_l, khw = self.grad.tm[name]
                    g_ptr, layout = raw.ptr, cin_l | (khw << 32)
                else:
                    g_ptr, layout = dict.__getitem__(self.grad, name).ptr, 0
                rows.append([w.ptr, self.mom[name].ptr, g_ptr, w.size, wd_bits | (block << 32), layout])
                block += (w.size + 1023) // 1024
            dev = self.ctx.empty((len(rows), 6), np.uint64)
            dev.copyfrom(np.array(rows, dtype=np.uint64))
            tab = self._sgd_table = (float(wd), dev, len(rows), block)
        lib.deepim_sgd_mom_update_multi(h, tab[1], tab[2], tab[3], c(lr), c(momentum), c(rescale_grad), c(clip_gradient or 0.0))
        orders = self._train_pack_orders()
        for name, shape in self.arg_shape_dict().items():
            if not name.endswith("_weight") or len(shape) != 4 or name.endswith("upsampling_weight"):
                continue
            base = name[: -len("_weight")]
            if base.startswith("deconv") or base.startswith("upsample_flow"):
                lib.deepim_deconv_pack_weights(h, self.packed[base], self.params[name], shape[0], shape[1])
            else:
                lib.deepim_conv_pack_weights_ex(h, self.packed[base], self.params[name], shape[0], shape[1], shape[2], shape[3],
                                                orders.get(base, 3))
        if self.B > self.FC6_PLAIN_MAX_BATCH:      # small batches read fc6's raw weights (_fc6)
            lib.deepim_fc_pack_weights(h, self.packed["fc6"], self.params["fc6_weight"], 256, 1024 * 8 * 10)

    def train_step(self, data, label, updater, iters=None, lr=None, wd=None, momentum=None, on_iter=None):
        """ONE training step as the reference runs it (deepim/core/module.py:1131-1137 with network.TRAIN_ITER_SIZE = 4, yaml
        :57-58): for every refinement iteration forward_backward → preds (rot_est, trans_est) → update (SGD + re-pack), and between
        iterations `interBatchUpdater.forward(data_batch, preds)` (lib/pair_matching/batch_updater_py_multi.py:91-328): pose ←
        RT_transform(src_pose, preds), re-render at it, new rot / trans labels (calc_RT_delta), K·T + lib/flow_c flow labels and
        weights, mask_rendered = depth > 0.2 — all resident, nothing allocated inside the loop, no host round trip.
        data must also carry tgt_pose [, depth_gt_observed with PRED_FLOW, class_index]; `updater` = batchUpdaterPyMulti with a
        device render machine. `on_iter(it, data, label)` is called after each iteration's update (tests, timers).
        Returns (data, label) of the last iteration."""
        t = self.cfg.TRAIN
        iters = int(iters or self.cfg.network.TRAIN_ITER_SIZE)
        lr = t.lr if lr is None else lr
        wd = t.wd if wd is None else wd
        momentum = t.momentum if momentum is None else momentum
        if getattr(self, "_upd_ws", None) is None or self._upd_ws[0] is not updater:
            self._upd_ws = (updater, updater.workspace(self.ctx, self.B))
        A = self.act
        for it in range(iters):
            self.forward_train(data, label)
            self.backward()
            preds = {"rot_est": A["rot_norm"], "trans_est": A["trans_est"]}     # get_outputs() before update(), as :1133-1134
            self.update(lr, wd, momentum)
            if on_iter is not None:
                on_iter(it, data, label)
            if it != iters - 1:
                batch = dict(label)
                batch.update(data)
                new = updater.forward(batch, preds, self.cfg, out=self._upd_ws[1])
                data = {k: new[k] for k in data}
                label = {k: new[k] for k in label}
        return data, label

    def _train_pack_orders(self):
        """layer → the ONE packed operand order its forward convolution reads in the training graph (NCHW activations,
        deepim_conv_weight_order): the per-step re-pack writes only that one."""
        order = lib.load().deepim_conv_weight_order      # evaluated per update: follows the context's conv options
        h, B = self.ctx.handle, self.B
        geo = {g[0]: g[1:] for g in self.enc_geom}
        geo.update({"Convolution1": (1024, 8, 10, 2, 3, 1, 1), "Convolution2": (1026, 15, 20, 2, 3, 1, 1),
                    "Convolution3": (770, 30, 40, 2, 3, 1, 1), "mask_conv3": (770, 30, 40, 1, 3, 1, 1)})
        return {n: order(h, B, cin, hh, ww, cout, k, k, s_, p_) for n, (cin, hh, ww, cout, k, s_, p_) in geo.items()}

    return dict(bind_train=bind_train, forward_train=forward_train, _dgrad=_dgrad,
                _small_conv_backward=_small_conv_backward, _head_conv_backward=_head_conv_backward,
                _deconv_backward=_deconv_backward, _decoder_backward=_decoder_backward, backward=backward, update=update, train_step=train_step,
                _train_pack_orders=_train_pack_orders)


for _name, _fn in _train_methods().items():
    setattr(deepIM_flownet, _name, _fn)
