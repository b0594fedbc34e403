"""ArcFace r50 (insightface `arcface_torch` IResNet-50) on MI355X -- the model behind ``meerqat.image.face_recognition``
(meerqat/image/face_recognition.py:55-61: ``get_model('r50', fp16=True)`` + ``backbone.pth``).

Every convolution is a GEMM on the split-bf16 matrix-pipe kernels (three bf16 MFMA products per fp32 product: fp32-class
accuracy, where the reference runs fp16 autocast); activations are NHWC, so a GEMM's output ``[B * Ho * Wo, Cout]`` IS the next
layer's input.  BatchNorms are eval-mode affines: the one BEHIND a convolution is folded into its weights and bias here, at load
(exact); the one IN FRONT of a convolution (IBasicBlock.bn1, the final bn2) and PReLU are elementwise operations on the tensor the
convolution reads.

The 48 3 x 3 convolutions of the blocks are IMPLICIT GEMMs (``mq_conv3x3_pair_f32``, csrc/conv.hip): no patch matrix is written,
the GEMM's LDS-DMA gathers the taps from the input's (hi, lo) pair, and the elementwise operations ride on the epilogue of the
convolution that PRODUCES a tensor.  An IBasicBlock is two launches:

    P1 = split(prelu(conv1(P0) + b2))                          # P0 = split(bn1(x)), written by the previous block
    x' = conv2(P1) + b3 + identity;   P0' = split(bn1'(x'))    # one epilogue: the fp32 shortcut of the next block + its input pair

(+ im2col and GEMM for the strided 1 x 1 downsample of a stage's first block).  The 3-channel stem (K = 27) is a direct fp32 convolution on
the vector ALU fused with its PReLU (``mq_stem_conv3x3_f32``: writes the first block's input pair and its downsample's operand, never
the full-resolution fp32 tensor); the head (bn2 - flatten - fc - features = one 7 x 7 "convolution", split-K) goes through
``mq_im2col_split_f32`` + the encoder GEMM; ``MQ_ARCFACE_CONV=im2col`` runs
EVERY convolution that way (round 4's first form).  The implicit kernel walks K as (32-channel block, tap) -- the nine taps of a block re-read the same
input lines back to back, +6 % from L2 hits alone; with ``MQ_CONV_KORDER=tap`` it walks K like the explicit path and the two
forwards agree bit for bit (tests/test_arcface_gpu.py), otherwise within fp32 rounding.

Parity: arcface_torch is not vendored by the reference and not installable here -- ``oracle/arcface.py`` restates the PUBLISHED
definition (parity unpinned); ``tests/test_arcface_gpu.py`` holds this module to that oracle within 1e-3."""
import os

import numpy as np
import torch

from . import _lib
from .encoders import EPI_BIAS, EPI_BIAS_RESIDUAL, SplitAct, _HipEncoder, _check_cuda, _stream, gemm_nt, split_bf16_tiled

EPS = 1e-5
LAYERS = (3, 4, 14, 3)


def _affine(state, name):
    """eval-mode BatchNorm ``name`` as (scale, shift) float64 arrays: y = x * scale + shift"""
    g, b = np.asarray(state[name + ".weight"], np.float64), np.asarray(state[name + ".bias"], np.float64)
    m, v = np.asarray(state[name + ".running_mean"], np.float64), np.asarray(state[name + ".running_var"], np.float64)
    a = g / np.sqrt(v + EPS)
    return a, b - m * a


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class _Conv:
    """One convolution as the GEMM sees it: W [Cout, Kpad] (kh, kw, cin order, the BatchNorm behind it folded in), bias [Cout]."""

    def __init__(self, weight, post_bn, k, stride, pad):
        w = np.asarray(weight, np.float64)                       # [Cout, Cin, k, k]
        a, b = post_bn
        w2 = w.transpose(0, 2, 3, 1).reshape(w.shape[0], -1) * a[:, None]
        self.k_true = w2.shape[1]
        self.kpad = (self.k_true + 31) // 32 * 32
        if self.kpad != self.k_true:
            w2 = np.concatenate([w2, np.zeros((w2.shape[0], self.kpad - self.k_true))], axis=1)
        self.w2, self.b = w2.astype(np.float32), b.astype(np.float32)
        self.cout, self.cin, self.k, self.stride, self.pad = w.shape[0], w.shape[1], k, stride, pad


class ArcFaceR50(_HipEncoder):
    """``model(pixel_values)`` -> fp32 [B, 512]; ``pixel_values`` fp32 [B, 3, 112, 112] in [-1, 1] on a GPU."""

    image_size = 112
    num_features = 512

    def __init__(self, state, layers=LAYERS, chunk=656):
        super().__init__()
        # chunk = faces per forward: the 14 x 14 stage (half the FLOPs) runs ceil(faces * 196 / 256) workgroups of one 256-row tile
        # each -- 328 faces = 252 of the 256 CUs in one round, 656 = 503 in two (256 faces: 196 of 256; measured 19.3 k -> 21.3 k
        # faces/s at 328 with the first implicit kernel; 23.8 k / 24.7 k at 328 / 656 with the final one)
        state = {k: _np(v).astype(np.float32) for k, v in state.items() if not k.endswith("num_batches_tracked")}
        self.layers, self.chunk = tuple(layers), int(chunk)
        self._convs, self._vecs = {}, {}

        def conv(name, post, k, stride, pad):
            c = _Conv(state[name + ".weight"], _affine(state, post), k, stride, pad)
            self._reg(name + ".w2", torch.from_numpy(c.w2))
            self._reg(name + ".b", torch.from_numpy(c.b))
            self._convs[name] = c

        def vec(name, arr):
            self._reg(name, torch.from_numpy(np.ascontiguousarray(arr, np.float32)))
            self._vecs[name] = name.replace(".", "_")

        conv("conv1", "bn1", 3, 1, 1)
        vec("prelu", state["prelu.weight"])
        # the stem as a direct convolution (mq_stem_conv3x3_f32): weights [27, 64], row (kh * 3 + kw) * 3 + c
        self._reg("conv1.wt", torch.from_numpy(np.ascontiguousarray(self._convs["conv1"].w2[:, :27].T)))
        for s, n in enumerate(self.layers, start=1):
            for i in range(n):
                p = f"layer{s}.{i}"
                a, b = _affine(state, p + ".bn1")
                vec(p + ".pre_scale", a)
                vec(p + ".pre_shift", b)
                conv(p + ".conv1", p + ".bn2", 3, 1, 1)
                vec(p + ".prelu", state[p + ".prelu.weight"])
                conv(p + ".conv2", p + ".bn3", 3, 2 if i == 0 else 1, 1)
                if i == 0:
                    conv(p + ".downsample.0", p + ".downsample.1", 1, 2, 0)
        # head: bn2 (on the im2col) - flatten in NCHW order - fc - BatchNorm1d.  The 7 x 7 x 512 patch of the NHWC activation has
        # column (h * 7 + w) * 512 + c where the checkpoint's fc has column c * 49 + h * 7 + w: permute once; fold `features` in.
        a2, b2 = _affine(state, "bn2")
        vec("head.pre_scale", a2)
        vec("head.pre_shift", b2)
        fa, fb = _affine(state, "features")
        wfc = np.asarray(state["fc.weight"], np.float64).reshape(-1, 512, 7, 7).transpose(0, 2, 3, 1).reshape(-1, 512 * 49)
        self._reg("fc.w2", torch.from_numpy((wfc * fa[:, None]).astype(np.float32)))
        self._reg("fc.b", torch.from_numpy((np.asarray(state["fc.bias"], np.float64) * fa + fb).astype(np.float32)))
        self.num_features = wfc.shape[0]

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_state_dict(cls, state, **kw):
        return cls(state, **kw)

    @classmethod
    def from_pretrained(cls, weight_path, **kw):
        """``backbone.pth`` of insightface's arcface_torch (a plain state dict), e.g. ms1mv3_arcface_r50_fp16/backbone.pth."""
        state = torch.load(os.fspath(weight_path), map_location="cpu", weights_only=True)
        return cls(state, **kw)

    # ------------------------------------------------------------------ forward
    def _v(self, name):
        return getattr(self, self._vecs[name]) if name is not None else None

    def _conv(self, name, x, B, H, W, nchw=False, slope=None, scale=None, shift=None, residual=None):
        """x: fp32 [B, H, W, Cin] (or NCHW) -> ([B * Ho * Wo, Cout] fp32, Ho, Wo)"""
        c = self._convs[name]
        Ho, Wo = (H + 2 * c.pad - c.k) // c.stride + 1, (W + 2 * c.pad - c.k) // c.stride + 1
        A = self._im2col(x, B, H, W, c.cin, nchw, c.k, c.k, c.stride, c.pad, c.kpad, slope, scale, shift)
        y = gemm_nt(A, getattr(self, (name + ".w2").replace(".", "_")), bias=getattr(self, (name + ".b").replace(".", "_")),
                    residual=residual, epilogue=EPI_BIAS_RESIDUAL if residual is not None else EPI_BIAS,
                    wsplit=self._wsplit(name + ".w2"))
        return y, Ho, Wo

    def _wsplit(self, name):
        w = getattr(self, name.replace(".", "_"))
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (name, w.device, w.data_ptr())
        if key not in cache:
            cache[key] = split_bf16_tiled(w)
        return cache[key]

    @staticmethod
    def _im2col(x, B, H, W, C, nchw, KH, KW, stride, pad, kpad, slope, scale, shift):
        lib = _lib.load()
        _check_cuda(x)
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        A = SplitAct.empty(B * Ho * Wo, kpad, x.device)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        with torch.cuda.device(x.device):
            _lib.check(lib.mq_im2col_split_f32(x.data_ptr(), B, H, W, C, int(nchw), KH, KW, stride, pad, p(slope), p(scale), p(shift),
                                               A.hi.data_ptr(), A.lo.data_ptr(), kpad, _stream(x)), "mq_im2col_split_f32")
        return A

    def _conv3x3(self, name, xin, B, H, W, slope=None, residual=None, scale=None, shift=None, tile=0):
        """The 3 x 3 convolution ``name`` as an implicit GEMM (mq_conv3x3_pair_f32) on ``xin``, the SplitAct [B * H * W, Cin] of its
        input AFTER the pre-operations.  ``slope``: -> SplitAct of prelu(conv + bias).  Otherwise: -> (fp32 conv + bias + residual,
        SplitAct of that * scale + shift or None)."""
        lib = _lib.load()
        c = self._convs[name]
        assert c.k == 3 and c.pad == 1 and xin.shape == (B * H * W, c.cin)
        Ho, Wo = (H - 1) // c.stride + 1, (W - 1) // c.stride + 1
        M, dev = B * Ho * Wo, xin.device
        if (tile == 0 and c.stride == 1 and W <= 127 and str(c.cout) in os.environ.get("MQ_CONV_PATCH", "64").split(",")
                and os.environ.get("MQ_CONV_KORDER", "channel") != "tap"):
            tile = 5  # MQ_CONV_TILE_PATCH_256x64: few output channels -- the nine taps read one LDS-resident input patch
        wh, wl = self._wsplit(name + ".w2")
        bias = getattr(self, (name + ".b").replace(".", "_"))
        zeros = self.__dict__.setdefault("_zero_pages", {})
        if dev not in zeros:
            zeros[dev] = torch.zeros(64, dtype=torch.uint8, device=dev)
        P = SplitAct.empty(M, c.cout, dev) if (slope is not None or scale is not None) else None
        Y = torch.empty((M, c.cout), dtype=torch.float32, device=dev) if slope is None else None
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, c.cin, c.stride, wh.data_ptr(), wl.data_ptr(),
                                               c.cout, bias.data_ptr(), p(slope), p(residual), p(scale), p(shift), p(Y),
                                               P.hi.data_ptr() if P is not None else None, P.lo.data_ptr() if P is not None else None,
                                               zeros[dev].data_ptr(), int(tile) | (0 if os.environ.get("MQ_CONV_KORDER", "channel") == "tap" else 0x100),
                                               _stream(xin.hi)), "mq_conv3x3_pair_f32")
        return (P if slope is not None else (Y, P)), Ho, Wo

    def _head(self, y, B, H, W):
        """bn2 - flatten - fc - features: one 7 x 7 "convolution".  [B, 512] outputs over K = 25,088: as one GEMM two workgroups would
        walk 784 K steps (1.65 ms of a 15-ms forward), so the K range is split over 49 workgroup rows and the fp32 partials summed
        (mq_gemm_nt_bf16x3s_splitk_f32; MQ_ARCFACE_HEAD_SPLITS=1: the one-pass GEMM)."""
        A = self._im2col(y, B, H, W, 512, False, H, W, 1, 0, H * W * 512, None, self._v("head.pre_scale"), self._v("head.pre_shift"))
        nsplit = int(os.environ.get("MQ_ARCFACE_HEAD_SPLITS", "49"))
        if nsplit <= 1:
            return gemm_nt(A, self.fc_w2, bias=self.fc_b, epilogue=EPI_BIAS, wsplit=self._wsplit("fc.w2"))
        lib = _lib.load()
        wh, wl = self._wsplit("fc.w2")
        M, K = A.shape
        N = self.fc_w2.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=y.device)
        part = torch.empty((nsplit, M, N), dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            _lib.check(lib.mq_gemm_nt_bf16x3s_splitk_f32(A.hi.data_ptr(), A.lo.data_ptr(), wh.data_ptr(), wl.data_ptr(), self.fc_b.data_ptr(),
                                                         out.data_ptr(), M, N, K, 1, nsplit, part.data_ptr(), _stream(y)),
                       "mq_gemm_nt_bf16x3s_splitk_f32")
        return out

    def forward(self, pixel_values):
        _check_cuda(pixel_values)
        x = pixel_values.to(torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise ValueError(f"expected [B, 3, {self.image_size}, {self.image_size}] pixel values, got {tuple(x.shape)}")
        if x.shape[0] > self.chunk:
            return torch.cat([self._forward(x[s:s + self.chunk]) for s in range(0, x.shape[0], self.chunk)])
        return self._forward(x)

    def _forward(self, x):
        if os.environ.get("MQ_ARCFACE_CONV", "implicit") == "im2col":
            return self._forward_im2col(x)
        B, H, W = x.shape[0], self.image_size, self.image_size
        blocks = [f"layer{s}.{i}" for s, n in enumerate(self.layers, start=1) for i in range(n)]
        # stem: direct fp32 convolution fused with its PReLU; writes the first block's conv1 input pair (bn1 applied) and the A operand
        # of its strided 1 x 1 downsample (the activated pixels at even coordinates) -- the full-resolution fp32 tensor is never stored
        lib = _lib.load()
        xin = SplitAct.empty(B * H * W, 64, x.device)
        y = None
        ds_in = SplitAct.empty(B * (H // 2) * (W // 2), 64, x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.mq_stem_conv3x3_f32(x.data_ptr(), B, H, W, self.conv1_wt.data_ptr(), self.conv1_b.data_ptr(), self._v("prelu").data_ptr(),
                                               self._v(blocks[0] + ".pre_scale").data_ptr(), self._v(blocks[0] + ".pre_shift").data_ptr(),
                                               xin.hi.data_ptr(), xin.lo.data_ptr(), ds_in.hi.data_ptr(), ds_in.lo.data_ptr(), _stream(x)),
                       "mq_stem_conv3x3_f32")
        for bi, p in enumerate(blocks):
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            o1, _, _ = self._conv3x3(p + ".conv1", xin, B, H, W, slope=self._v(p + ".prelu"))
            if (p + ".downsample.0") in self._convs:
                if ds_in is not None:   # the first block: the stem wrote the 1 x 1 stride-2 "patches"
                    name = p + ".downsample.0"
                    identity = gemm_nt(ds_in, getattr(self, (name + ".w2").replace(".", "_")), bias=getattr(self, (name + ".b").replace(".", "_")),
                                       epilogue=EPI_BIAS, wsplit=self._wsplit(name + ".w2"))
                    ds_in = None
                else:
                    identity, _, _ = self._conv(p + ".downsample.0", y, B, H, W)
            else:
                identity = y
            (y, xin), H, W = self._conv3x3(p + ".conv2", o1, B, H, W, residual=identity,
                                           scale=self._v(nxt + ".pre_scale") if nxt else None,
                                           shift=self._v(nxt + ".pre_shift") if nxt else None)
        return self._head(y, B, H, W)

    def _forward_im2col(self, x):
        """Round 4's first form, kept as the check of the implicit one (MQ_ARCFACE_CONV=im2col): every convolution an explicit
        im2col + GEMM.  Same products in the same order: the two forwards agree bit for bit."""
        B, H, W = x.shape[0], self.image_size, self.image_size
        # stem: conv - BN (folded); its PReLU is applied where the result is read (the first block's im2cols)
        y, H, W = self._conv("conv1", x, B, H, W, nchw=True)
        pending = self._v("prelu")
        for s, n in enumerate(self.layers, start=1):
            for i in range(n):
                p = f"layer{s}.{i}"
                o1, _, _ = self._conv(p + ".conv1", y, B, H, W, slope=pending, scale=self._v(p + ".pre_scale"), shift=self._v(p + ".pre_shift"))
                if i == 0:
                    identity, Ho, Wo = self._conv(p + ".downsample.0", y, B, H, W, slope=pending)
                else:
                    assert pending is None
                    identity = y
                y, Ho, Wo = self._conv(p + ".conv2", o1, B, H, W, slope=self._v(p + ".prelu"), residual=identity)
                H, W, pending = Ho, Wo, None
        return self._head(y, B, H, W)
