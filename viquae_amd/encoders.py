"""HIP-backed encoders with the Hugging Face call surface the reference uses.

``get_pretrained(class_name, ...)`` in the reference resolves ``DPRContextEncoder``,
``DPRQuestionEncoder`` and ``CLIPModel`` in ``transformers`` (meerqat/data/loading.py:167-183); the
mirror in :mod:`viquae_amd.data.loading` resolves the same names here first.  The objects below are
``torch.nn.Module``s (so ``model.to(device).eval()`` and the reference's ``nn.DataParallel`` wrapping,
meerqat/ir/embedding.py:285-288, keep working) whose ``forward`` runs entirely in libmeerqat_hip.so
(csrc/encoder.hip): fused embedding+LayerNorm, MFMA GEMMs with fused bias/GELU/residual epilogues,
attention, LayerNorm.  Weights are read from a Hugging Face checkpoint directory (``config.json`` +
``model.safetensors`` or ``pytorch_model.bin``) or from a ``state_dict`` with HF tensor names.

fp32 in, fp32 out; parity target <= 1e-3 abs against the HF implementations (tests/golden/).
No CPU fallback: calling ``forward`` without the HIP library or a GPU raises.
"""
import json
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL = 0, 1, 2, 3, 4


class ModelOutput(dict):
    """dict with attribute access: ``outputs['pooler_output']`` (meerqat/ir/embedding.py:231-234) and
    ``outputs.pooler_output`` both work."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


# --------------------------------------------------------------------------------------------------
# thin op wrappers over the C ABI (torch tensors in, torch tensors out, current stream)
# --------------------------------------------------------------------------------------------------
def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MeerqatHipError("viquae_amd encoders run on MI355X only: move the model and its inputs to a GPU "
                                       "(no CPU fallback)")


def split_bf16(w):
    """(hi, lo) bf16 halves of an fp32 CUDA tensor, as int16 tensors: w ~= hi + lo to 2^-18 relative."""
    _check_cuda(w)
    lib = _lib.load()
    w = w.contiguous()
    hi = torch.empty(w.shape, dtype=torch.int16, device=w.device)
    lo = torch.empty(w.shape, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mq_split_bf16_f32(w.data_ptr(), w.numel(), hi.data_ptr(), lo.data_ptr(), _stream(w)), "mq_split_bf16_f32")
    return hi, lo


GEMM_W_TILED = 0x100  # include/meerqat_hip.h MQ_GEMM_W_TILED


class TiledSplit(tuple):
    """(hi, lo) of :func:`split_bf16_tiled`: a weight's bf16 split in the GEMM kernels' tile layout."""
    tiled = True


def split_bf16_tiled(w):
    """The split of :func:`split_bf16` for a weight matrix [N, K] (K % 32 == 0), stored tile by tile
    ([ceil(N / 256)][K / 32][256][32], zero rows beyond N): the operand of one K step of a GEMM tile is one contiguous block.
    ``gemm_nt(..., wsplit=TiledSplit)`` gives the same bits as with the row-major split, 3.5-6 % sooner."""
    _check_cuda(w)
    lib = _lib.load()
    w = w.contiguous()
    N, K = w.shape
    n = int(lib.mq_split_bf16_tiled_elems(N, K))
    hi = torch.empty(n, dtype=torch.int16, device=w.device)
    lo = torch.empty(n, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mq_split_bf16_tiled_f32(w.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), _stream(w)), "mq_split_bf16_tiled_f32")
    return TiledSplit((hi, lo))


def _tile(rm):
    """row-major [M, K] int16 -> the pair layout ([ceil(M / 256)][K / 32][256][32], flat)"""
    M, K = rm.shape
    Mt = (M + 255) // 256
    if Mt * 256 != M:
        rm = torch.cat([rm, rm.new_zeros((Mt * 256 - M, K))])
    return rm.view(Mt, 256, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(-1)


def _untile(flat, M, K):
    """the pair layout -> row-major [M, K]"""
    Mt = (M + 255) // 256
    return flat.view(Mt, K // 32, 256, 32).permute(0, 2, 1, 3).reshape(Mt * 256, K)[:M]


class SplitAct:
    """An activation [rows, features] kept as its (hi, lo) bf16 pair: what the kernels producing GEMM inputs write in
    split_bf16 mode, and what ``mq_gemm_nt_bf16x3s_f32`` consumes.  ``hi`` / ``lo`` are flat int16 tensors in the library's PAIR
    LAYOUT (include/meerqat_hip.h: tile by tile, [rows / 256][features / 32][256][32], rows padded to 256);
    :meth:`rowmajor` gives the plain [rows, features] view of either."""
    __slots__ = ("hi", "lo", "shape")

    def __init__(self, hi, lo, shape=None):
        if shape is None:  # row-major [M, K] tensors (e.g. from split_bf16): convert
            shape = tuple(hi.shape)
            hi, lo = _tile(hi), _tile(lo)
        self.hi, self.lo, self.shape = hi, lo, (int(shape[0]), int(shape[1]))

    @classmethod
    def empty(cls, rows, features, device):
        if features % 32:
            raise ValueError("split activations need a multiple of 32 features")
        n = ((rows + 255) // 256) * 256 * features
        return cls(torch.empty(n, dtype=torch.int16, device=device), torch.empty(n, dtype=torch.int16, device=device),
                   (rows, features))

    @property
    def device(self):
        return self.hi.device

    def rowmajor(self):
        """(hi, lo) as plain [rows, features] int16 tensors"""
        return _untile(self.hi, *self.shape), _untile(self.lo, *self.shape)

    def first_rows(self, B, L):
        """rows 0, L, 2L, ... (the first token of each of B sequences)"""
        return self.rows(torch.arange(0, B * L, L, device=self.hi.device))

    def rows(self, index):
        """the rows `index` (int64 tensor)"""
        K = self.shape[1]
        blk, r = index // 256, index % 256
        hi = self.hi.view(-1, K // 32, 256, 32)[blk, :, r, :].reshape(-1, K)
        lo = self.lo.view(-1, K // 32, 256, 32)[blk, :, r, :].reshape(-1, K)
        return SplitAct(hi, lo)

    def float(self):
        hi, lo = self.rowmajor()
        return hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()


def _use_split(*feature_counts):
    """split activations are used when every GEMM depth of the block is a multiple of the kernel's K step"""
    return _gemm_mode() == "split_bf16" and all(k % 32 == 0 for k in feature_counts)


def gemm_nt(a, w, bias=None, residual=None, epilogue=EPI_NONE, out=None, wsplit=None, out_split=False):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T).  ``wsplit`` = (hi, lo) from :func:`split_bf16` selects the
    split-bf16 kernels (fp32-class accuracy on the bf16 matrix pipe; K must be a multiple of 32).  ``a`` may be a
    :class:`SplitAct` (no conversions in the GEMM loop); ``out_split`` returns the result as a :class:`SplitAct`."""
    lib = _lib.load()
    if isinstance(a, SplitAct):
        _check_cuda(a.hi, w)
        if wsplit is None:
            raise ValueError("a split activation needs split weights")
        M, K = a.shape
        N = w.shape[0]
        b = bias.data_ptr() if bias is not None else None
        res = SplitAct.empty(M, N, a.device) if out_split else (
            out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device))
        flags = GEMM_W_TILED if getattr(wsplit, "tiled", False) else 0
        if isinstance(residual, SplitAct):  # the shortcut read from a pair (hi + lo): no fp32 copy of that tensor needed
            if residual.shape != (M, N):
                raise ValueError(f"residual pair of shape {residual.shape} for a [{M}, {N}] product")
            with torch.cuda.device(a.device):
                _lib.check(lib.mq_gemm_nt_bf16x3s_respair_f32(
                    a.hi.data_ptr(), a.lo.data_ptr(), wsplit[0].data_ptr(), wsplit[1].data_ptr(), b, residual.hi.data_ptr(),
                    residual.lo.data_ptr(), None if out_split else res.data_ptr(), res.hi.data_ptr() if out_split else None,
                    res.lo.data_ptr() if out_split else None, M, N, K, epilogue | flags, _stream(a.hi)), "mq_gemm_nt_bf16x3s_respair_f32")
            return res
        r = residual.data_ptr() if residual is not None else None
        with torch.cuda.device(a.device):
            _lib.check(lib.mq_gemm_nt_bf16x3s_f32(
                a.hi.data_ptr(), a.lo.data_ptr(), wsplit[0].data_ptr(), wsplit[1].data_ptr(), b, r,
                None if out_split else res.data_ptr(), res.hi.data_ptr() if out_split else None,
                res.lo.data_ptr() if out_split else None, M, N, K, epilogue | flags, _stream(a.hi)), "mq_gemm_nt_bf16x3s_f32")
        return res
    if out_split:
        raise ValueError("out_split needs a split input activation")
    _check_cuda(a, w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    b = bias.data_ptr() if bias is not None else None
    r = residual.data_ptr() if residual is not None else None
    with torch.cuda.device(a.device):
        if wsplit is not None and K % 32 == 0:
            flags = GEMM_W_TILED if getattr(wsplit, "tiled", False) else 0
            _lib.check(lib.mq_gemm_nt_bf16x3_f32(a.data_ptr(), wsplit[0].data_ptr(), wsplit[1].data_ptr(), b, r, out.data_ptr(),
                                                 M, N, K, epilogue | flags, _stream(a)), "mq_gemm_nt_bf16x3_f32")
        else:
            _lib.check(lib.mq_gemm_nt_f32(a.data_ptr(), w.data_ptr(), b, r, out.data_ptr(), M, N, K, epilogue, _stream(a)),
                       "mq_gemm_nt_f32")
    return out


def _gemm_mode():
    """MQ_ENC_GEMM = split_bf16 (default) | f32"""
    return os.environ.get("MQ_ENC_GEMM", "split_bf16")


def _pair_residual():
    """MQ_ENC_RESIDUAL = pair (default) | f32: where the post-LayerNorm shortcuts of the BERT stacks are read from"""
    return os.environ.get("MQ_ENC_RESIDUAL", "pair") != "f32"


def layernorm(x, g, b, eps, out=None):
    _check_cuda(x)
    lib = _lib.load()
    M, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.mq_layernorm_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), M, C, float(eps), _stream(x)),
                   "mq_layernorm_f32")
    return out


def layernorm_split(x, g, b, eps, f32_out=None, want_f32=True):
    """LayerNorm writing the split pair of its output and, when ``want_f32``, the fp32 output too (into ``f32_out`` if
    given).  -> (fp32 tensor or None, SplitAct)."""
    _check_cuda(x)
    lib = _lib.load()
    M, C = x.shape
    y = (f32_out if f32_out is not None else torch.empty_like(x)) if want_f32 else None
    sp = SplitAct.empty(M, C, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mq_layernorm_split_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr() if y is not None else None,
                                              sp.hi.data_ptr(), sp.lo.data_ptr(), M, C, float(eps), _stream(x)),
                   "mq_layernorm_split_f32")
    return y, sp


def attention(qkv, mask, B, L, heads, scale, causal=False, split=False, bf16x3=None):
    """softmax(q k^T * scale + mask) v.  ``bf16x3`` (default: on in split_bf16 mode) computes both products as
    three-term split-bf16 sums on the bf16 matrix pipe; ``split`` returns the output as a :class:`SplitAct`."""
    _check_cuda(qkv)
    lib = _lib.load()
    H = qkv.shape[1] // 3
    if bf16x3 is None:
        bf16x3 = _gemm_mode() == "split_bf16"
    sp = SplitAct.empty(B * L, H, qkv.device) if split else None
    out = None if split else torch.empty((B * L, H), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(lib.mq_attention_split_f32(qkv.data_ptr(), mask.data_ptr() if mask is not None else None,
                                              out.data_ptr() if out is not None else None,
                                              sp.hi.data_ptr() if split else None, sp.lo.data_ptr() if split else None, B, L,
                                              heads, H // heads, float(scale), int(bool(causal)), int(bool(bf16x3)),
                                              _stream(qkv)), "mq_attention_split_f32")
    return sp if split else out


def attention_packed(qkv, cu_seqlens, classes, heads, scale, causal=False, split=False, bf16x3=None):
    """Attention over a PACKED token matrix (sequence b = rows [cu[b], cu[b+1]), every key real).  ``classes`` =
    [(int32 sequence ids on the device, longest length among them)]: one launch per length class."""
    _check_cuda(qkv)
    lib = _lib.load()
    T, H = qkv.shape[0], qkv.shape[1] // 3
    if bf16x3 is None:
        bf16x3 = _gemm_mode() == "split_bf16"
    sp = SplitAct.empty(T, H, qkv.device) if split else None
    out = None if split else torch.empty((T, H), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        for seq_ids, max_len in classes:
            _lib.check(lib.mq_attention_packed_f32(qkv.data_ptr(), cu_seqlens.data_ptr(), seq_ids.data_ptr(), seq_ids.numel(),
                                                   int(max_len), out.data_ptr() if out is not None else None,
                                                   sp.hi.data_ptr() if split else None, sp.lo.data_ptr() if split else None,
                                                   heads, H // heads, float(scale), int(bool(causal)), int(bool(bf16x3)),
                                                   _stream(qkv)), "mq_attention_packed_f32")
    return sp if split else out


# --------------------------------------------------------------------------------------------------
# checkpoint reading
# --------------------------------------------------------------------------------------------------
def read_checkpoint(path):
    """(config dict, state dict of CPU tensors) from a Hugging Face ``save_pretrained`` directory."""
    with open(os.path.join(path, "config.json")) as f:
        config = json.load(f)
    st = os.path.join(path, "model.safetensors")
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(st):
        from safetensors.torch import load_file
        state = load_file(st)
    elif os.path.exists(pt):
        state = torch.load(pt, map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"no model.safetensors or pytorch_model.bin under {path}")
    return config, state


def _t(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.detach().to(torch.float32).contiguous()


class _HipEncoder(nn.Module):
    def _reg(self, name, tensor):
        self.register_buffer(name.replace(".", "_"), _t(tensor), persistent=False)
        return getattr(self, name.replace(".", "_"))

    def _ws(self, name):
        """(hi, lo) bf16 split of weight buffer `name`, built once per device (None in f32 mode)."""
        if _gemm_mode() != "split_bf16":
            return None
        w = getattr(self, name)
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (name, w.device, w.data_ptr())
        if key not in cache:  # tile layout whenever the split-bf16 kernels can take the weight at all (K % 32 == 0)
            tiled = w.dim() == 2 and w.shape[1] % 32 == 0 and os.environ.get("MQ_ENC_W_TILED", "1") != "0"
            cache[key] = split_bf16_tiled(w) if tiled else split_bf16(w)
        return cache[key]


# --------------------------------------------------------------------------------------------------
# BERT / DPR
# --------------------------------------------------------------------------------------------------
class BertEncoderHIP(_HipEncoder):
    """BertModel without pooler (DPR builds it with add_pooling_layer=False)."""

    def __init__(self, config, state, prefix):
        super().__init__()
        g = lambda n: state[prefix + n]  # noqa: E731
        self.hidden = int(config["hidden_size"])
        self.layers = int(config["num_hidden_layers"])
        self.heads = int(config["num_attention_heads"])
        self.eps = float(config.get("layer_norm_eps", 1e-12))
        if config.get("hidden_act", "gelu") != "gelu":
            raise NotImplementedError("only the exact-erf GELU of bert-base is provided")
        if config.get("position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("only absolute position embeddings")
        if self.hidden // self.heads != 64:
            raise NotImplementedError("attention head size must be 64")
        self._reg("w_word", g("embeddings.word_embeddings.weight"))
        self._reg("w_pos", g("embeddings.position_embeddings.weight"))
        self._reg("w_type", g("embeddings.token_type_embeddings.weight"))
        self._reg("emb_g", g("embeddings.LayerNorm.weight"))
        self._reg("emb_b", g("embeddings.LayerNorm.bias"))
        for i in range(self.layers):
            p = f"encoder.layer.{i}."
            self._reg(f"l{i}_wqkv", torch.cat([_t(g(p + f"attention.self.{n}.weight")) for n in ("query", "key", "value")]))
            self._reg(f"l{i}_bqkv", torch.cat([_t(g(p + f"attention.self.{n}.bias")) for n in ("query", "key", "value")]))
            self._reg(f"l{i}_wo", g(p + "attention.output.dense.weight"))
            self._reg(f"l{i}_bo", g(p + "attention.output.dense.bias"))
            self._reg(f"l{i}_g1", g(p + "attention.output.LayerNorm.weight"))
            self._reg(f"l{i}_b1", g(p + "attention.output.LayerNorm.bias"))
            self._reg(f"l{i}_wi", g(p + "intermediate.dense.weight"))
            self._reg(f"l{i}_bi", g(p + "intermediate.dense.bias"))
            self._reg(f"l{i}_w2", g(p + "output.dense.weight"))
            self._reg(f"l{i}_b2", g(p + "output.dense.bias"))
            self._reg(f"l{i}_g2", g(p + "output.LayerNorm.weight"))
            self._reg(f"l{i}_b2n", g(p + "output.LayerNorm.bias"))

    @torch.no_grad()
    def forward(self, input_ids, token_type_ids=None, attention_mask=None, output_hidden_states=False, cls_only=False):
        cls_only = cls_only and not output_hidden_states
        _check_cuda(input_ids, self.w_word)
        lib = _lib.load()
        B, L = input_ids.shape
        if L > self.w_pos.shape[0]:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {self.w_pos.shape[0]}")
        dev = input_ids.device
        ids = input_ids.to(torch.int64).contiguous()
        tt = token_type_ids.to(torch.int64).contiguous() if token_type_ids is not None else None
        mask = attention_mask.to(torch.int64).contiguous() if attention_mask is not None else None
        H = self.hidden
        split = self.uses_split()  # activations that only feed GEMMs travel as (hi, lo) bf16 pairs
        h = torch.empty((B * L, H), dtype=torch.float32, device=dev)
        hs = SplitAct.empty(B * L, H, dev) if split else None
        with torch.cuda.device(dev):
            _lib.check(lib.mq_bert_embed_ln_split_f32(
                ids.data_ptr(), tt.data_ptr() if tt is not None else None, self.w_word.data_ptr(), self.w_pos.data_ptr(),
                self.w_type.data_ptr(), self.emb_g.data_ptr(), self.emb_b.data_ptr(), h.data_ptr(),
                hs.hi.data_ptr() if split else None, hs.lo.data_ptr() if split else None, B, L, H, self.eps, _stream(h)),
                "mq_bert_embed_ln_split_f32")
        return self._layers(h, hs, mask, B, L, output_hidden_states, cls_only)

    def uses_split(self):
        return _use_split(self.hidden, self.l0_wi.shape[0])

    @torch.no_grad()
    def forward_packed(self, input_ids, token_type_ids, pack):
        """[CLS] vectors [B, H] of a right-padded batch through the PACKED forward: only the real tokens of all sequences
        (``pack`` from :func:`_pack_plan`) go through the embedding, the GEMMs and the LayerNorms; attention runs per
        sequence over exactly its own keys.  Bit-identical to the dense forward's [CLS] rows (see _pack_plan)."""
        _check_cuda(input_ids, self.w_word)
        lib = _lib.load()
        dev = input_ids.device
        keep, pos, cu, classes, cls_rows = pack
        T, H = int(keep.numel()), self.hidden
        ids = input_ids.to(torch.int64).reshape(-1).index_select(0, keep)
        tt = token_type_ids.to(torch.int64).reshape(-1).index_select(0, keep) if token_type_ids is not None else None
        split = self.uses_split()
        h = torch.empty((T, H), dtype=torch.float32, device=dev)
        hs = SplitAct.empty(T, H, dev) if split else None
        with torch.cuda.device(dev):
            _lib.check(lib.mq_bert_embed_ln_packed_f32(
                ids.data_ptr(), tt.data_ptr() if tt is not None else None, pos.data_ptr(), self.w_word.data_ptr(),
                self.w_pos.data_ptr(), self.w_type.data_ptr(), self.emb_g.data_ptr(), self.emb_b.data_ptr(), h.data_ptr(),
                hs.hi.data_ptr() if split else None, hs.lo.data_ptr() if split else None, T, H, self.eps, _stream(h)),
                "mq_bert_embed_ln_packed_f32")
        return self.packed_layers(h, hs, cu, classes, cls_rows)

    @torch.no_grad()
    def packed_layers(self, h, hs, cu, classes, cls_rows):
        """The encoder stack over PACKED embeddings h [T, H] (hs = their split pair in split mode, or None to make it here):
        sequence b = rows [cu[b], cu[b + 1]), attention per sequence (``classes`` of :func:`pack_plan_from_lengths`) -> the
        rows ``cls_rows`` of the last layer's output, [B, H]."""
        H = self.hidden
        split = self.uses_split()
        if split and hs is None:
            hs = SplitAct(*split_bf16_tiled(h), shape=h.shape)
        scale = 1.0 / math.sqrt(H // self.heads)
        for i in range(self.layers):
            w = lambda n: getattr(self, f"l{i}_{n}")  # noqa: E731
            sp = lambda n: self._ws(f"l{i}_{n}")  # noqa: E731
            last = i == self.layers - 1
            if split:
                qkv = gemm_nt(hs, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
                ctx = attention_packed(qkv, cu, classes, self.heads, scale, split=True)
                pair_res = _pair_residual()  # see _layers
                if last:  # everything after the last attention is row-wise: only the [CLS] rows go on
                    ctx, h = ctx.rows(cls_rows), (h.index_select(0, cls_rows).contiguous() if h is not None else None)
                    hs = hs.rows(cls_rows)
                a = gemm_nt(ctx, w("wo"), w("bo"), hs if pair_res else h, EPI_BIAS_RESIDUAL, wsplit=sp("wo"))
                h1, h1s = layernorm_split(a, w("g1"), w("b1"), self.eps, f32_out=a, want_f32=not pair_res)
                f = gemm_nt(h1s, w("wi"), w("bi"), None, EPI_BIAS_GELU, wsplit=sp("wi"), out_split=True)
                o = gemm_nt(f, w("w2"), w("b2"), h1s if pair_res else h1, EPI_BIAS_RESIDUAL, wsplit=sp("w2"))
                h, hs = layernorm_split(o, w("g2"), w("b2n"), self.eps, f32_out=o, want_f32=not pair_res or last)
            else:
                qkv = gemm_nt(h, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
                ctx = attention_packed(qkv, cu, classes, self.heads, scale)
                if last:
                    ctx, h = ctx.index_select(0, cls_rows).contiguous(), h.index_select(0, cls_rows).contiguous()
                a = gemm_nt(ctx, w("wo"), w("bo"), h, EPI_BIAS_RESIDUAL, wsplit=sp("wo"))
                h1 = layernorm(a, w("g1"), w("b1"), self.eps, out=a)
                f = gemm_nt(h1, w("wi"), w("bi"), None, EPI_BIAS_GELU, wsplit=sp("wi"))
                o = gemm_nt(f, w("w2"), w("b2"), h1, EPI_BIAS_RESIDUAL, wsplit=sp("w2"))
                h = layernorm(o, w("g2"), w("b2n"), self.eps, out=o)
        return h

    @torch.no_grad()
    def _layers(self, h, hs, mask, B, L, output_hidden_states=False, cls_only=False):
        """The encoder stack over embeddings h [B*L, H] (hs = their split pair in split mode, or None to make it here)
        with an int64 0/1 mask [B, L] -> (last hidden [B, L or 1, H], hidden states or None)."""
        H = self.hidden
        split = self.uses_split()
        if split and hs is None:
            hs = SplitAct(*split_bf16_tiled(h), shape=h.shape)  # the tile layout of the weights IS the pair layout
        hidden = [h.view(B, L, H)] if output_hidden_states else None
        scale = 1.0 / math.sqrt(H // self.heads)
        for i in range(self.layers):
            w = lambda n: getattr(self, f"l{i}_{n}")  # noqa: E731
            sp = lambda n: self._ws(f"l{i}_{n}")  # noqa: E731
            last_cls = cls_only and i == self.layers - 1
            if split:
                qkv = gemm_nt(hs, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
                ctx = attention(qkv, mask, B, L, self.heads, scale, split=True)
                if last_cls:
                    # DPR reads only last_hidden_state[:, 0]: after the last attention, the output projection, both
                    # LayerNorms and the FFN are row-wise, so run them on the [CLS] rows alone (same numbers, 1/L of the rows)
                    ctx = ctx.first_rows(B, L)
                    h = h.view(B, L, H)[:, 0, :].contiguous() if h is not None else None
                    hs = hs.first_rows(B, L)
                # Shortcuts read the LayerNorm outputs from their (hi, lo) pairs (value hi + lo: 16 mantissa bits, 2^-17 relative),
                # so a LayerNorm writes the pair only -- a third less traffic per LayerNorm (MQ_ENC_RESIDUAL=f32: the fp32 copies)
                pair_res = _pair_residual()
                a = gemm_nt(ctx, w("wo"), w("bo"), hs if pair_res else h, EPI_BIAS_RESIDUAL, wsplit=sp("wo"))
                h1, h1s = layernorm_split(a, w("g1"), w("b1"), self.eps, f32_out=a, want_f32=not pair_res)
                f = gemm_nt(h1s, w("wi"), w("bi"), None, EPI_BIAS_GELU, wsplit=sp("wi"), out_split=True)
                o = gemm_nt(f, w("w2"), w("b2"), h1s if pair_res else h1, EPI_BIAS_RESIDUAL, wsplit=sp("w2"))
                need_f32 = not pair_res or output_hidden_states or i == self.layers - 1
                h, hs = layernorm_split(o, w("g2"), w("b2n"), self.eps, f32_out=o, want_f32=need_f32)
            else:
                qkv = gemm_nt(h, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
                ctx = attention(qkv, mask, B, L, self.heads, scale)
                if last_cls:
                    ctx = ctx.view(B, L, H)[:, 0, :].contiguous()
                    h = h.view(B, L, H)[:, 0, :].contiguous()
                a = gemm_nt(ctx, w("wo"), w("bo"), h, EPI_BIAS_RESIDUAL, wsplit=sp("wo"))
                h1 = layernorm(a, w("g1"), w("b1"), self.eps, out=a)
                f = gemm_nt(h1, w("wi"), w("bi"), None, EPI_BIAS_GELU, wsplit=sp("wi"))
                o = gemm_nt(f, w("w2"), w("b2"), h1, EPI_BIAS_RESIDUAL, wsplit=sp("w2"))
                h = layernorm(o, w("g2"), w("b2n"), self.eps, out=o)
            if output_hidden_states:
                hidden.append(h.view(B, L, H))
        if cls_only:
            return h.view(B, 1, H), hidden
        return h.view(B, L, H), hidden


_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _length_buckets(attention_mask, max_buckets=8):
    """[(int64 sequence indices on the device, length)] for a right-padded 0/1 mask with padding worth skipping, else
    None (no mask, other mask shapes, empty sequences, nothing to gain, or MQ_ENC_PAD_SKIP=0).  One small D2H copy."""
    if attention_mask is None or os.environ.get("MQ_ENC_PAD_SKIP", "1") == "0":
        return None
    B, L = attention_mask.shape
    if B == 0 or L < 2:
        return None
    m = attention_mask != 0
    right_padded = bool((m[:, 1:] <= m[:, :-1]).all()) and bool(((attention_mask == 0) | (attention_mask == 1)).all())
    if not right_padded:
        return None
    lens = m.sum(dim=1).cpu().numpy()
    if lens.min() < 1 or int(lens.sum()) > 0.9 * B * L:
        return None
    # a group must still fill the chip (its GEMMs have real_tokens / 256 row tiles): ~32 k tokens each, else the dense
    # forward of a small batch is already one wave of workgroups and splitting it only adds launches
    nb = int(max(1, min(max_buckets, lens.sum() // 32768)))
    if nb == 1 and int(lens.max()) > 0.9 * L:
        return None  # one group at (nearly) the padded length: nothing to gain
    order = np.argsort(lens, kind="stable")
    plan = []
    for part in _partition_by_length(lens[order], nb):
        seqs = order[part]
        plan.append((torch.from_numpy(np.ascontiguousarray(seqs)).to(attention_mask.device), int(lens[seqs].max())))
    return plan


def _pack_plan(attention_mask):
    """(kept flat token indices, position id per kept token, cu_seqlens, attention launch classes, [CLS] rows) for a
    right-padded 0/1 mask with padding worth skipping -- else None (no mask, other mask shapes, empty sequences,
    < 10 % padding, or MQ_ENC_PACKED=0).  One small D2H copy (the lengths).

    Packed forward: the reference pads every passage to max_length (256, experiments/ir/viquae/dpr/passages/config.json:11-14)
    although a 100-word passage has ~130 tokens.  Every operation of the encoder is row-wise except attention, and in
    attention a padded key contributes exactly 0 to every sum and never sets a row maximum: dropping the padded ROWS
    everywhere and the padded KEYS in attention leaves the [CLS] vectors bit-identical to the dense forward."""
    if attention_mask is None or os.environ.get("MQ_ENC_PACKED", "1") == "0" or os.environ.get("MQ_ENC_PAD_SKIP", "1") == "0":
        return None
    B, L = attention_mask.shape
    if B == 0 or L < 2:
        return None
    m = attention_mask != 0
    right_padded = bool((m[:, 1:] <= m[:, :-1]).all()) and bool(((attention_mask == 0) | (attention_mask == 1)).all())
    if not right_padded:
        return None
    lens = m.sum(dim=1).cpu().numpy()
    return pack_plan_from_lengths(lens, L, attention_mask.device)


def pack_plan_from_lengths(lens, L, device, stage=None):
    """The plan of :func:`_pack_plan` from the sequence lengths alone (HOST int array [B]; the mask is right-padded 0/1 by
    construction, e.g. lengths the tokenizer reported): every array is built on the host and copied to ``device`` on the
    CURRENT stream -- no device -> host synchronisation, so an input pipeline can prepare the plan of batch i + 1 on a side
    stream while batch i runs (viquae_amd/pipeline.py).  ``stage`` (optional): numpy array -> page-locked CPU tensor holding a
    copy of it; the copies to the device are then truly asynchronous (a copy from pageable memory blocks the calling thread,
    on this runtime until the device has drained).  None when packing does not pay (see _pack_plan)."""
    if os.environ.get("MQ_ENC_PACKED", "1") == "0" or os.environ.get("MQ_ENC_PAD_SKIP", "1") == "0":
        return None
    lens = np.asarray(lens, dtype=np.int64)
    B = lens.shape[0]
    if B == 0 or L < 2 or lens.min() < 1 or lens.max() > L or int(lens.sum()) > 0.9 * B * L:
        return None
    cu_host = np.concatenate([[0], np.cumsum(lens)])
    T = int(cu_host[-1])
    # flat indices b * L + t of the real tokens, in row-major order (= nonzero() of the flattened mask)
    pos_host = np.arange(T, dtype=np.int64) - np.repeat(cu_host[:-1], lens)
    keep_host = pos_host + np.repeat(np.arange(B, dtype=np.int64) * L, lens)

    def dev(a):
        a = np.ascontiguousarray(a)
        return (stage(a) if stage is not None else torch.from_numpy(a)).to(device, non_blocking=True)

    keep, pos, cu = dev(keep_host), dev(pos_host.astype(np.int32)), dev(cu_host.astype(np.int32))
    classes = []
    for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 1 << 30)):               # the attention kernel's key-tile counts
        sel = np.nonzero((lens > lo) & (lens <= hi))[0].astype(np.int32)
        if sel.size:
            classes.append((dev(sel), int(lens[sel].max())))
    cls_rows = dev(cu_host[:-1].astype(np.int64))
    return keep, pos, cu, classes, cls_rows


def _partition_by_length(sorted_lens, nb):
    """Cuts ascending lengths into <= nb contiguous groups minimising sum(group size x group's longest length), the
    tokens a dense forward of each group processes (dynamic programme over the distinct lengths) -> list of slices."""
    u, cnt = np.unique(sorted_lens, return_counts=True)
    U = len(u)
    nb = max(1, min(nb, U))
    cs = np.concatenate([[0], np.cumsum(cnt)])
    cost = np.full((nb + 1, U + 1), np.inf)
    cost[0, 0] = 0.0
    arg = np.zeros((nb + 1, U + 1), dtype=np.int64)
    for b in range(1, nb + 1):
        for j in range(1, U + 1):
            c = cost[b - 1, :j] + (cs[j] - cs[:j]) * float(u[j - 1])  # last group = distinct lengths i .. j-1
            i = int(np.argmin(c))
            cost[b, j], arg[b, j] = c[i], i
    b = int(np.argmin(cost[1:, U])) + 1
    cuts, j = [], U
    while b > 0:
        i = int(arg[b, j])
        cuts.append(slice(int(cs[i]), int(cs[j])))
        j, b = i, b - 1
    return [c for c in reversed(cuts) if c.stop > c.start]


class _DPREncoder(_HipEncoder):
    _prefix = None
    config_class = dict

    def __init__(self, config, state):
        super().__init__()
        self.config = dict(config)
        if int(self.config.get("projection_dim", 0) or 0) != 0:
            raise NotImplementedError("DPR projection_dim > 0 is not used by the reference checkpoints")
        self.bert_model = BertEncoderHIP(self.config, state, self._prefix)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        config, state = read_checkpoint(pretrained_model_name_or_path)
        return cls(config, state)

    @classmethod
    def from_state_dict(cls, config, state):
        return cls(config, state)

    supports_pack_plan = True  # forward(..., pack_plan=pack_plan_from_lengths(...)): see viquae_amd/pipeline.py

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, output_hidden_states=False, pack_plan=None,
                **unused):
        """``pack_plan`` (optional): the packed-forward plan of THIS batch prepared ahead of time from the tokenizer's
        lengths (:func:`pack_plan_from_lengths`) -- the caller vouches that ``attention_mask`` is the right-padded mask of
        those lengths; the forward then needs no device -> host copy at all."""
        pack = None if output_hidden_states else (pack_plan if pack_plan is not None else _pack_plan(attention_mask))
        if pack is not None:
            return ModelOutput(pooler_output=self.bert_model.forward_packed(input_ids, token_type_ids, pack))
        plan = None if output_hidden_states else _length_buckets(attention_mask)
        if plan is not None:
            return ModelOutput(pooler_output=self._forward_buckets(plan, input_ids, attention_mask, token_type_ids))
        last, hidden = self.bert_model(input_ids, token_type_ids, attention_mask, output_hidden_states, cls_only=True)
        out = ModelOutput(pooler_output=last[:, 0, :].contiguous())
        if output_hidden_states:
            out["hidden_states"] = tuple(hidden)
        return out

    def _forward_buckets(self, plan, input_ids, attention_mask, token_type_ids):
        return _pooled_by_groups(self.bert_model, plan, input_ids, attention_mask, token_type_ids)


def _pooled_by_groups(bert, plan, input_ids, attention_mask, token_type_ids=None, embeddings=None):
    """Padding-aware forward: the batch is cut into a few groups of similar length, each run dense at ITS longest
    length.  The reference pads every passage to 256 tokens (experiments/ir/viquae/dpr/passages/config.json:11-14:
    ``padding: max_length``) although a 100-word passage has ~130: about half of the dense work is padding.  Every
    operation is row-wise except attention, where a masked key contributes exactly 0 to every sum and never to the
    row maximum -- so the pooled [CLS] vectors are bit-identical to the dense forward (asserted in the tests).
    ``embeddings`` [B, L, H] (ECA: text + face + image tokens) replaces the token ids when given."""
    ref = embeddings if embeddings is not None else input_ids
    B, dev = ref.shape[0], ref.device
    out = torch.empty((B, bert.hidden), dtype=torch.float32, device=dev)
    # groups alternate between two side streams: the tail of one group's GEMM (a partial wave of workgroups) is
    # filled by the other group's kernels
    main = torch.cuda.current_stream(dev)
    streams = _side_streams(dev, min(int(os.environ.get("MQ_ENC_GROUP_STREAMS", "2")), len(plan)))
    ready = torch.cuda.Event()
    ready.record(main)

    def run(idx, Li):
        mask = attention_mask.index_select(0, idx)[:, :Li].contiguous()
        if embeddings is not None:
            h = embeddings.index_select(0, idx)[:, :Li].contiguous()
            last, _ = bert._layers(h.view(-1, bert.hidden), None, mask, idx.numel(), Li, False, cls_only=True)
        else:
            ids = input_ids.index_select(0, idx)[:, :Li]
            tt = token_type_ids.index_select(0, idx)[:, :Li] if token_type_ids is not None else None
            last, _ = bert(ids, tt, mask, False, cls_only=True)
        out.index_copy_(0, idx, last[:, 0, :])

    first = 0
    warm_key = (str(dev), bert.w_word.data_ptr(), _gemm_mode())
    if bert.__dict__.get("_weights_split") != warm_key:
        # the (hi, lo) weight splits are built lazily by the first forward: build them on the caller's stream before
        # two other streams start reading them
        run(*plan[0])
        first = 1
        ready.record(main)
        bert.__dict__["_weights_split"] = warm_key
    for n, (idx, Li) in enumerate(plan[first:]):
        st = streams[n % len(streams)]
        st.wait_event(ready)
        with torch.cuda.stream(st):
            run(idx, Li)
    for st in streams:
        main.wait_stream(st)
    return out


class DPRContextEncoder(_DPREncoder):
    """transformers.DPRContextEncoder (experiments/ir/viquae/dpr/passages/config.json:2-5)."""
    _prefix = "ctx_encoder.bert_model."


class DPRQuestionEncoder(_DPREncoder):
    """transformers.DPRQuestionEncoder (experiments/ir/viquae/dpr/questions/config.json)."""
    _prefix = "question_encoder.bert_model."


# --------------------------------------------------------------------------------------------------
# CLIP vision tower
# --------------------------------------------------------------------------------------------------
def _clip_block(h, w, sp, mask, B, T, heads, scale, eps, act, causal, keep_rows=None, pack=None):
    """One pre-LN CLIP transformer block on the residual stream h [B*T, H] (updated in place).  ``keep_rows`` (int64 row
    indices, one per sequence) is given for the LAST block: only the pooled token's row leaves the tower, and everything
    after the attention is row-wise, so the output projection and the MLP run on those B rows alone (same numbers).
    ``pack`` = (cu_seqlens, attention launch classes) when h is a PACKED token matrix (see _pack_plan)."""
    def attend(qkv, split):
        if pack is not None:
            return attention_packed(qkv, pack[0], pack[1], heads, scale, causal=causal, split=split)
        return attention(qkv, mask, B, T, heads, scale, causal=causal, split=split)

    if _use_split(h.shape[1], w("w1").shape[0]):
        _, y = layernorm_split(h, w("g1"), w("b1"), eps, want_f32=False)
        qkv = gemm_nt(y, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
        ctx = attend(qkv, True)
        if keep_rows is not None:
            ctx, h = ctx.rows(keep_rows), h.index_select(0, keep_rows).contiguous()
        h = gemm_nt(ctx, w("wo"), w("bo"), h, EPI_BIAS_RESIDUAL, out=h, wsplit=sp("wo"))
        _, y = layernorm_split(h, w("g2"), w("b2n"), eps, want_f32=False)
        f = gemm_nt(y, w("w1"), w("bb1"), None, act, wsplit=sp("w1"), out_split=True)
        return gemm_nt(f, w("w2"), w("bb2"), h, EPI_BIAS_RESIDUAL, out=h, wsplit=sp("w2"))
    y = layernorm(h, w("g1"), w("b1"), eps)
    qkv = gemm_nt(y, w("wqkv"), w("bqkv"), None, EPI_BIAS, wsplit=sp("wqkv"))
    ctx = attend(qkv, False)
    if keep_rows is not None:
        ctx, h = ctx.index_select(0, keep_rows).contiguous(), h.index_select(0, keep_rows).contiguous()
    h = gemm_nt(ctx, w("wo"), w("bo"), h, EPI_BIAS_RESIDUAL, out=h, wsplit=sp("wo"))
    y = layernorm(h, w("g2"), w("b2n"), eps)
    f = gemm_nt(y, w("w1"), w("bb1"), None, act, wsplit=sp("w1"))
    return gemm_nt(f, w("w2"), w("bb2"), h, EPI_BIAS_RESIDUAL, out=h, wsplit=sp("w2"))


class CLIPModel(_HipEncoder):
    """transformers.CLIPModel restricted to the two calls the reference makes: ``get_image_features``
    (experiments/image_embedding/clip/vit_config.json:18: ViT vision tower + visual projection) and
    ``get_text_features`` (experiments/ir/viquae/clip/config.json:15: causal text tower, EOT pooling, text projection).
    Either tower may be absent from the checkpoint; calling the missing one raises."""
    config_class = dict

    def __init__(self, config, state):
        super().__init__()
        self.config = dict(config)
        self.has_vision = "vision_model.embeddings.class_embedding" in state
        self.has_text = "text_model.embeddings.token_embedding.weight" in state
        if not (self.has_vision or self.has_text):
            raise ValueError("the checkpoint holds neither a CLIP vision tower nor a CLIP text tower")
        if self.has_text:
            self._init_text(config, state)
        if not self.has_vision:
            return
        v = dict(config.get("vision_config", config))
        self.hidden = int(v["hidden_size"])
        self.layers = int(v["num_hidden_layers"])
        self.heads = int(v["num_attention_heads"])
        self.image_size, self.patch = int(v["image_size"]), int(v["patch_size"])
        self.channels = int(v.get("num_channels", 3))
        self.eps = float(v.get("layer_norm_eps", 1e-5))
        act = v.get("hidden_act", "quick_gelu")
        if act not in ("quick_gelu", "gelu"):
            raise NotImplementedError(f"CLIP activation {act}")
        self.act = EPI_BIAS_QUICKGELU if act == "quick_gelu" else EPI_BIAS_GELU
        if self.hidden // self.heads != 64:
            raise NotImplementedError("attention head size must be 64")
        s = state
        self._reg("cls", s["vision_model.embeddings.class_embedding"])
        self._reg("wpe", _t(s["vision_model.embeddings.patch_embedding.weight"]).reshape(self.hidden, -1))
        self._reg("pos", s["vision_model.embeddings.position_embedding.weight"])
        self._reg("pre_g", s["vision_model.pre_layrnorm.weight"])
        self._reg("pre_b", s["vision_model.pre_layrnorm.bias"])
        self._reg("post_g", s["vision_model.post_layernorm.weight"])
        self._reg("post_b", s["vision_model.post_layernorm.bias"])
        self._reg("wproj", s["visual_projection.weight"])
        for i in range(self.layers):
            p = f"vision_model.encoder.layers.{i}."
            self._reg(f"l{i}_wqkv", torch.cat([_t(s[p + f"self_attn.{n}.weight"]) for n in ("q_proj", "k_proj", "v_proj")]))
            self._reg(f"l{i}_bqkv", torch.cat([_t(s[p + f"self_attn.{n}.bias"]) for n in ("q_proj", "k_proj", "v_proj")]))
            self._reg(f"l{i}_wo", s[p + "self_attn.out_proj.weight"])
            self._reg(f"l{i}_bo", s[p + "self_attn.out_proj.bias"])
            self._reg(f"l{i}_g1", s[p + "layer_norm1.weight"])
            self._reg(f"l{i}_b1", s[p + "layer_norm1.bias"])
            self._reg(f"l{i}_g2", s[p + "layer_norm2.weight"])
            self._reg(f"l{i}_b2n", s[p + "layer_norm2.bias"])
            self._reg(f"l{i}_w1", s[p + "mlp.fc1.weight"])
            self._reg(f"l{i}_bb1", s[p + "mlp.fc1.bias"])
            self._reg(f"l{i}_w2", s[p + "mlp.fc2.weight"])
            self._reg(f"l{i}_bb2", s[p + "mlp.fc2.bias"])

    def _init_text(self, config, state):
        t = dict(config.get("text_config", config))
        self.t_hidden = int(t["hidden_size"])
        self.t_layers = int(t["num_hidden_layers"])
        self.t_heads = int(t["num_attention_heads"])
        self.t_eps = float(t.get("layer_norm_eps", 1e-5))
        self.t_eos = int(t.get("eos_token_id", 2))
        self.t_max_pos = int(t.get("max_position_embeddings", 77))
        act = t.get("hidden_act", "quick_gelu")
        if act not in ("quick_gelu", "gelu"):
            raise NotImplementedError(f"CLIP activation {act}")
        self.t_act = EPI_BIAS_QUICKGELU if act == "quick_gelu" else EPI_BIAS_GELU
        if self.t_hidden // self.t_heads != 64:
            raise NotImplementedError("attention head size must be 64")
        s = state
        self._reg("t_tok", s["text_model.embeddings.token_embedding.weight"])
        self._reg("t_pos", s["text_model.embeddings.position_embedding.weight"])
        self._reg("t_fin_g", s["text_model.final_layer_norm.weight"])
        self._reg("t_fin_b", s["text_model.final_layer_norm.bias"])
        self._reg("t_wproj", s["text_projection.weight"])
        for i in range(self.t_layers):
            p = f"text_model.encoder.layers.{i}."
            self._reg(f"t{i}_wqkv", torch.cat([_t(s[p + f"self_attn.{n}.weight"]) for n in ("q_proj", "k_proj", "v_proj")]))
            self._reg(f"t{i}_bqkv", torch.cat([_t(s[p + f"self_attn.{n}.bias"]) for n in ("q_proj", "k_proj", "v_proj")]))
            self._reg(f"t{i}_wo", s[p + "self_attn.out_proj.weight"])
            self._reg(f"t{i}_bo", s[p + "self_attn.out_proj.bias"])
            self._reg(f"t{i}_g1", s[p + "layer_norm1.weight"])
            self._reg(f"t{i}_b1", s[p + "layer_norm1.bias"])
            self._reg(f"t{i}_g2", s[p + "layer_norm2.weight"])
            self._reg(f"t{i}_b2n", s[p + "layer_norm2.bias"])
            self._reg(f"t{i}_w1", s[p + "mlp.fc1.weight"])
            self._reg(f"t{i}_bb1", s[p + "mlp.fc1.bias"])
            self._reg(f"t{i}_w2", s[p + "mlp.fc2.weight"])
            self._reg(f"t{i}_bb2", s[p + "mlp.fc2.bias"])

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        config, state = read_checkpoint(pretrained_model_name_or_path)
        return cls(config, state)

    @classmethod
    def from_state_dict(cls, config, state):
        return cls(config, state)

    def forward(self, *a, **k):
        raise NotImplementedError("only get_image_features / get_text_features are on the reference's path "
                                  "(experiments/image_embedding/clip/vit_config.json:18, experiments/ir/viquae/clip/config.json:15)")

    @torch.no_grad()
    def get_text_features(self, input_ids=None, attention_mask=None, position_ids=None, **unused):
        """-> f32 [B, projection_dim] tensor, like ``get_image_features`` and like the transformers releases the reference
        pins (its config has no ``output_key``; ``embed`` ignores ``output_key`` for tensor outputs)."""
        if not self.has_text:
            raise NotImplementedError("this checkpoint has no text tower (text_model.* tensors)")
        if position_ids is not None:
            raise NotImplementedError("explicit position_ids")
        _check_cuda(input_ids, self.t_tok)
        lib = _lib.load()
        ids = input_ids.to(torch.int64).contiguous()
        if ids.dim() != 2:
            raise ValueError("input_ids must be [batch, length]")
        B, L = ids.shape
        if L > self.t_max_pos:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: {L} "
                             f"and max_position_embeddings: {self.t_max_pos}")
        if B and (int(ids.min()) < 0 or int(ids.max()) >= self.t_tok.shape[0]):
            raise IndexError("input_ids outside the vocabulary")
        eot = ids.to(torch.int32).argmax(dim=1) if self.t_eos == 2 else (ids == self.t_eos).to(torch.int32).argmax(dim=1)
        pack = _pack_plan(attention_mask)
        if pack is not None and bool((eot.to(torch.int64) < attention_mask.to(torch.int64).sum(dim=1)).all()):
            return self._text_features_packed(ids, eot, pack)
        plan = _length_buckets(attention_mask)
        if plan is not None:
            # titles padded to the longest of 2048 (experiments/ir/viquae/clip/config.json:10-13) are mostly padding: run
            # groups of similar length at their own length -- the pooled end-of-text row only attends to real tokens, so
            # the features are bit-identical to the dense forward
            out = torch.empty((B, self.t_wproj.shape[0]), dtype=torch.float32, device=ids.device)
            for idx, Li in plan:
                out.index_copy_(0, idx, self._text_features_dense(ids.index_select(0, idx)[:, :Li].contiguous(),
                                                                  attention_mask.index_select(0, idx)[:, :Li]))
            return out
        return self._text_features_dense(ids, attention_mask)

    def _text_features_packed(self, ids, eot, pack):
        """Packed forward of right-padded titles (``pack`` from :func:`_pack_plan`): only the real tokens go through the
        tower and causal attention runs per title over its own keys.  The pooled end-of-text row ``eot[b]`` (a real token)
        attends to real tokens only, and so does every row it depends on: bit-identical to the dense forward."""
        lib = _lib.load()
        keep, pos, cu, classes, first_rows = pack
        dev, H, T = ids.device, self.t_hidden, int(keep.numel())
        ids_p = ids.reshape(-1).index_select(0, keep)
        h = torch.empty((T, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mq_clip_text_embed_packed_f32(ids_p.data_ptr(), pos.data_ptr(), self.t_tok.data_ptr(),
                                                         self.t_pos.data_ptr(), h.data_ptr(), T, H, _stream(h)),
                       "mq_clip_text_embed_packed_f32")
        scale = (H // self.t_heads) ** -0.5
        eot_rows = first_rows + eot.to(torch.int64)
        for i in range(self.t_layers):
            w = lambda n: getattr(self, f"t{i}_{n}")  # noqa: E731
            sp = lambda n: self._ws(f"t{i}_{n}")  # noqa: E731
            h = _clip_block(h, w, sp, None, 0, 0, self.t_heads, scale, self.t_eps, self.t_act, causal=True,
                            keep_rows=eot_rows if i == self.t_layers - 1 else None, pack=(cu, classes))
        pooled = layernorm(h, self.t_fin_g, self.t_fin_b, self.t_eps)
        return gemm_nt(pooled, self.t_wproj, None, None, EPI_NONE, wsplit=self._ws("t_wproj"))

    def _text_features_dense(self, ids, attention_mask):
        lib = _lib.load()
        B, L = ids.shape
        mask = attention_mask.to(torch.int64).contiguous() if attention_mask is not None else None
        dev, H = ids.device, self.t_hidden
        h = torch.empty((B * L, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mq_clip_text_embed_f32(ids.data_ptr(), self.t_tok.data_ptr(), self.t_pos.data_ptr(), h.data_ptr(),
                                                  B, L, H, _stream(h)), "mq_clip_text_embed_f32")
        scale = (H // self.t_heads) ** -0.5
        for i in range(self.t_layers):
            w = lambda n: getattr(self, f"t{i}_{n}")  # noqa: E731
            sp = lambda n: self._ws(f"t{i}_{n}")  # noqa: E731
            keep = None
            if i == self.t_layers - 1:
                # end-of-text position per sequence (HF: the largest id for legacy eos_token_id == 2 configs, else the first
                # eos_token_id): only that row of the last block's output is pooled
                eot = ids.to(torch.int32).argmax(dim=1) if self.t_eos == 2 else (ids == self.t_eos).to(torch.int32).argmax(dim=1)
                keep = torch.arange(B, device=dev, dtype=torch.int64) * L + eot.to(torch.int64)
            h = _clip_block(h, w, sp, mask, B, L, self.t_heads, scale, self.t_eps, self.t_act, causal=True, keep_rows=keep)
        pooled = layernorm(h, self.t_fin_g, self.t_fin_b, self.t_eps)  # h = the B end-of-text rows after the last block
        return gemm_nt(pooled, self.t_wproj, None, None, EPI_NONE, wsplit=self._ws("t_wproj"))

    @torch.no_grad()
    def get_image_features(self, pixel_values=None, **unused):
        if not self.has_vision:
            raise NotImplementedError("this checkpoint has no vision tower (vision_model.* tensors)")
        _check_cuda(pixel_values, self.cls)
        lib = _lib.load()
        px = pixel_values.to(torch.float32).contiguous()
        B, C, S, S2 = px.shape
        if (C, S, S2) != (self.channels, self.image_size, self.image_size):
            raise ValueError(f"expected pixel_values [B,{self.channels},{self.image_size},{self.image_size}], got {tuple(px.shape)}")
        dev, H = px.device, self.hidden
        G = S // self.patch
        T = G * G + 1
        patches = torch.empty((B * G * G, C * self.patch * self.patch), dtype=torch.float32, device=dev)
        h = torch.empty((B * T, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mq_clip_patchify_f32(px.data_ptr(), patches.data_ptr(), B, C, S, self.patch, _stream(px)),
                       "mq_clip_patchify_f32")
            pe = gemm_nt(patches, self.wpe, None, None, EPI_NONE, wsplit=self._ws("wpe"))
            _lib.check(lib.mq_clip_assemble_ln_f32(pe.data_ptr(), self.cls.data_ptr(), self.pos.data_ptr(), self.pre_g.data_ptr(),
                                                   self.pre_b.data_ptr(), h.data_ptr(), B, T, H, self.eps, _stream(px)),
                       "mq_clip_assemble_ln_f32")
        scale = (H // self.heads) ** -0.5
        for i in range(self.layers):
            w = lambda n: getattr(self, f"l{i}_{n}")  # noqa: E731
            sp = lambda n: self._ws(f"l{i}_{n}")  # noqa: E731
            last = i == self.layers - 1
            h = _clip_block(h, w, sp, None, B, T, self.heads, scale, self.eps, self.act, causal=False,
                            keep_rows=torch.arange(B, device=dev, dtype=torch.int64) * T if last else None)
        pooled = layernorm(h, self.post_g, self.post_b, self.eps)  # h = the B [CLS] rows after the last block
        return gemm_nt(pooled, self.wproj, None, None, EPI_NONE, wsplit=self._ws("wproj"))



# --------------------------------------------------------------------------------------------------
# multimodal encoders (meerqat/models/mm.py): ECAEncoder (:557-754), IntermediateLinearFusion (:773-861)
# --------------------------------------------------------------------------------------------------
class MMConfig:
    """The fields of the reference's MMConfig / ILFConfig (meerqat/models/mm.py:20-87,757-770) this build reads;
    ``meerqat.ir.embedding.is_multimodal`` (:147-152) recognises the model by this class name."""

    def __init__(self, config):
        c = dict(config)
        self.raw = c
        self.hidden_size = int(c["hidden_size"])
        self.layer_norm_eps = float(c.get("layer_norm_eps", 1e-12))
        self.n_images = int(c.get("n_images", 1))
        self.n_faces = int(c.get("n_faces", 4))
        self.face_kwargs = dict(c.get("face_kwargs") or dict(face_dim=512, bbox_dim=7))
        self.image_kwargs = dict(c.get("image_kwargs") or {"clip-RN50": {"input_dim": 1024}, "imagenet-RN50": {"input_dim": 2048}})
        self.face_and_image_are_exclusive = bool(c.get("face_and_image_are_exclusive", False))
        self.no_text = bool(c.get("no_text", False))
        self.gating = bool(c.get("gating", False))
        self.question_encoder = bool(c.get("question_encoder", True))
        if self.n_images != 1:
            raise NotImplementedError("n_images > 1 (image type embeddings) is outside this build; the shipped configs use 1")


def _pad_k(x, multiple=32):
    """zero-pad the feature dimension of a [rows, K] matrix to a multiple of the GEMM's K step"""
    K = x.shape[1]
    Kp = -(-K // multiple) * multiple
    if Kp == K:
        return x.contiguous()
    out = torch.zeros((x.shape[0], Kp), dtype=torch.float32, device=x.device)
    out[:, :K] = x
    return out


class _MMEmbeddings(_HipEncoder):
    """FaceEmbedding / ImageEmbedding of meerqat/models/image.py as GEMMs with fused bias / residual epilogues.
    A tanh gate (mm.py:610-629, meerqat/models/utils.py:11-27) multiplies a module's output by the scalar
    tanh(gate_param): it is folded into the LayerNorm's (faces) or the linear layer's (images) parameters at load time."""

    def _init_mm(self, mmc, state, gated):
        self.mm = mmc
        self.config = mmc  # what is_multimodal() and get_inputs() look at
        H = mmc.hidden_size
        if mmc.n_faces > 0:
            g = math.tanh(float(_t(state["face_gate.gate_param"]).reshape(-1)[0])) if gated else 1.0
            self._reg("f_w", _pad_k(_t(state["face_embedding.face_proj.weight"])))
            self._reg("f_b", state["face_embedding.face_proj.bias"])
            self._reg("b_w", _pad_k(_t(state["face_embedding.bbox_proj.weight"])))
            self._reg("b_b", state["face_embedding.bbox_proj.bias"])
            self._reg("f_g", _t(state["face_embedding.LayerNorm.weight"]) * g)
            self._reg("f_beta", _t(state["face_embedding.LayerNorm.bias"]) * g)
        self.image_names = list(mmc.image_kwargs)
        for n, name in enumerate(self.image_names):
            g = math.tanh(float(_t(state[f"image_gates.{name}.gate_param"]).reshape(-1)[0])) if gated else 1.0
            self._reg(f"i{n}_w", _pad_k(_t(state[f"image_embeddings.{name}.linear.weight"]) * g))
            self._reg(f"i{n}_b", _t(state[f"image_embeddings.{name}.linear.bias"]) * g)
        assert H % 2 == 0

    def _faces(self, face_inputs, B):
        """-> (face embeddings fp32 [B * n_faces, H] or None, face mask int64 [B, n_faces])"""
        nf = self.mm.n_faces
        fmask = face_inputs["attention_mask"].reshape(B, -1).to(torch.int64)
        if nf == 0:
            return None, fmask
        face = _pad_k(face_inputs["face"].to(torch.float32).reshape(B * nf, -1))
        bbox = _pad_k(face_inputs["bbox"].to(torch.float32).reshape(B * nf, -1))
        e = gemm_nt(face, self.f_w, self.f_b, None, EPI_BIAS, wsplit=self._ws("f_w"))
        e = gemm_nt(bbox, self.b_w, self.b_b, e, EPI_BIAS_RESIDUAL, out=e, wsplit=self._ws("b_w"))
        return layernorm(e, self.f_g, self.f_beta, self.mm.layer_norm_eps, out=e), fmask

    def _image(self, n, x, residual=None):
        x = _pad_k(x.to(torch.float32))
        if residual is None:
            return gemm_nt(x, getattr(self, f"i{n}_w"), getattr(self, f"i{n}_b"), None, EPI_BIAS, wsplit=self._ws(f"i{n}_w"))
        return gemm_nt(x, getattr(self, f"i{n}_w"), getattr(self, f"i{n}_b"), residual, EPI_BIAS_RESIDUAL, out=residual,
                       wsplit=self._ws(f"i{n}_w"))


class ECAEncoder(_MMEmbeddings):
    """meerqat.models.mm.ECAEncoder: text, face and image tokens concatenated at the sequence level, BERT encoder,
    [CLS] vector.  Call surface of the reference: ``model(text_inputs=..., face_inputs=..., image_inputs=...)``."""
    config_class = dict

    def __init__(self, config, state):
        super().__init__()
        mmc = MMConfig(config)
        self.bert_model = BertEncoderHIP(config, state, "bert_model.")
        self._init_mm(mmc, state, mmc.gating)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        config, state = read_checkpoint(pretrained_model_name_or_path)
        return cls(config, state)

    @classmethod
    def from_state_dict(cls, config, state):
        return cls(config, state)

    @torch.no_grad()
    def forward(self, text_inputs=None, face_inputs=None, image_inputs=None, output_attentions=False,
                output_hidden_states=False, return_dict=True, **unused):
        if output_attentions:
            raise NotImplementedError("attention maps are not materialised by the fused attention kernel")
        bert, mmc = self.bert_model, self.mm
        ids = text_inputs["input_ids"]
        _check_cuda(ids, bert.w_word)
        lib = _lib.load()
        mask = text_inputs["attention_mask"].to(torch.int64)
        tt = text_inputs.get("token_type_ids")
        if mmc.no_text:  # only the [CLS] token of the text (mm.py:724-729)
            ids, mask = ids[:, :1], mask[:, :1]
            tt = tt[:, :1] if tt is not None else None
        ids = ids.to(torch.int64).contiguous()
        tt = tt.to(torch.int64).contiguous() if tt is not None else None
        B, L = ids.shape
        dev, H = ids.device, bert.hidden
        te = torch.empty((B * L, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mq_bert_embed_ln_split_f32(
                ids.data_ptr(), tt.data_ptr() if tt is not None else None, bert.w_word.data_ptr(), bert.w_pos.data_ptr(),
                bert.w_type.data_ptr(), bert.emb_g.data_ptr(), bert.emb_b.data_ptr(), te.data_ptr(), None, None, B, L, H,
                bert.eps, _stream(te)), "mq_bert_embed_ln_split_f32")
        parts, masks = [te.view(B, L, H)], [mask]
        fo, fmask = self._faces(face_inputs, B)
        if fo is not None:
            parts.append(fo.view(B, mmc.n_faces, H))
        masks.append(fmask)
        imasks = []
        for n, name in enumerate(self.image_names):
            image = image_inputs[name]
            parts.append(self._image(n, image["input"].reshape(B, -1)).view(B, 1, H))
            imasks.append(image["attention_mask"].reshape(B, 1).to(torch.int64))
        if imasks:
            imask = torch.cat(imasks, dim=1)
            if mmc.face_and_image_are_exclusive:  # mask the images of examples with at least one face (mm.py:716-720)
                imask = imask * (fmask.sum(dim=1, keepdim=True) == 0).to(torch.int64)
            masks.append(imask)
        h = torch.cat(parts, dim=1).contiguous()
        full_mask = torch.cat(masks, dim=1).contiguous()
        Lt = h.shape[1]
        if not output_hidden_states and return_dict and os.environ.get("MQ_ENC_PAD_SKIP", "1") != "0" and bool((full_mask[:, 0] != 0).all()):
            # The text is padded to max_length in the MIDDLE of the joint sequence (text | faces | images).  The layers are
            # permutation-equivariant (positions were added by the embeddings), so the attended tokens are moved to the
            # front in their original order ([CLS] stays row 0) and the padding-aware forward applies.  Masked keys add
            # exact zeros, but the surviving keys sit in other MFMA k-groups: equal to the dense forward up to fp32
            # summation order (~1e-6), not bit for bit.
            order = torch.argsort((full_mask == 0).to(torch.int8), dim=1, stable=True)
            cmask = torch.gather(full_mask, 1, order)
            pack = _pack_plan(cmask)
            if pack is not None:
                # the PACKED forward over the joint sequences (round 3): only the attended tokens of every example go through
                # the stack, attention runs per example over exactly its own keys (12.5 k -> see bench.py eca_multimodal_encoder)
                keep, _, cu, classes, cls_rows = pack
                flat = (order + torch.arange(B, device=dev)[:, None] * Lt).reshape(-1).index_select(0, keep)
                pooled = bert.packed_layers(h.view(B * Lt, H).index_select(0, flat), None, cu, classes, cls_rows)
                return ModelOutput(pooler_output=pooled, last_hidden_state=pooled[:, None, :], hidden_states=None, attentions=None)
            plan = _length_buckets(cmask)
            if plan is not None:
                hc = torch.gather(h, 1, order[:, :, None].expand(-1, -1, H))
                pooled = _pooled_by_groups(bert, plan, None, cmask, embeddings=hc)
                return ModelOutput(pooler_output=pooled, last_hidden_state=pooled[:, None, :], hidden_states=None, attentions=None)
        last, hidden = bert._layers(h.view(B * Lt, H), None, full_mask, B, Lt, output_hidden_states, cls_only=not output_hidden_states)
        pooled = last[:, 0, :]
        if not return_dict:
            return (pooled, last) + ((tuple(hidden),) if hidden is not None else ())
        return ModelOutput(pooler_output=pooled, last_hidden_state=last,
                           hidden_states=tuple(hidden) if hidden is not None else None, attentions=None)


class IntermediateLinearFusion(_MMEmbeddings):
    """meerqat.models.mm.IntermediateLinearFusion: LayerNorm(dpr_proj(DPR [CLS]) + sum of the face embeddings + the image
    projections).  Like the reference, ALL n_faces slots are summed, padded ones included (mm.py:838-843)."""
    config_class = dict

    def __init__(self, config, state):
        super().__init__()
        mmc = MMConfig(config)
        prefix = "dpr_encoder.question_encoder.bert_model." if mmc.question_encoder else "dpr_encoder.ctx_encoder.bert_model."
        self.bert_model = BertEncoderHIP(config, state, prefix)
        self._init_mm(mmc, state, False)
        self._reg("p_w", state["dpr_proj.weight"])
        self._reg("p_b", state["dpr_proj.bias"])
        self._reg("ln_g", state["LayerNorm.weight"])
        self._reg("ln_b", state["LayerNorm.bias"])

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        config, state = read_checkpoint(pretrained_model_name_or_path)
        return cls(config, state)

    @classmethod
    def from_state_dict(cls, config, state):
        return cls(config, state)

    @torch.no_grad()
    def forward(self, text_inputs=None, face_inputs=None, image_inputs=None, **unused):
        mmc = self.mm
        lib = _lib.load()
        pack = _pack_plan(text_inputs.get("attention_mask"))
        plan = None if pack is not None else _length_buckets(text_inputs.get("attention_mask"))
        if pack is not None:     # the DPR encoders' packed forward: real tokens only, bit-identical [CLS] rows
            pooled = self.bert_model.forward_packed(text_inputs["input_ids"], text_inputs.get("token_type_ids"), pack)
        elif plan is not None:
            pooled = _pooled_by_groups(self.bert_model, plan, text_inputs["input_ids"], text_inputs["attention_mask"],
                                       text_inputs.get("token_type_ids"))
        else:
            last, _ = self.bert_model(text_inputs["input_ids"], text_inputs.get("token_type_ids"),
                                      text_inputs.get("attention_mask"), cls_only=True)
            pooled = last[:, 0, :].contiguous()
        B, H = pooled.shape
        out = gemm_nt(pooled, self.p_w, self.p_b, None, EPI_BIAS, wsplit=self._ws("p_w"))
        fo, fmask = self._faces(face_inputs, B)
        if fo is not None:
            with torch.cuda.device(out.device):
                _lib.check(lib.mq_sum_groups_f32(fo.data_ptr(), out.data_ptr(), out.data_ptr(), B, mmc.n_faces, H, _stream(out)),
                           "mq_sum_groups_f32")
        for n, name in enumerate(self.image_names):
            x = image_inputs[name]["input"].reshape(B, -1).to(torch.float32)
            if mmc.face_and_image_are_exclusive:  # zero the image features of examples with a detected face (mm.py:852-856)
                x = x * (fmask.sum(dim=1, keepdim=True) == 0).to(torch.float32)
            out = self._image(n, x, residual=out)
        return ModelOutput(pooler_output=layernorm(out, self.ln_g, self.ln_b, mmc.layer_norm_eps, out=out))


HIP_CLASSES = {"DPRContextEncoder": DPRContextEncoder, "DPRQuestionEncoder": DPRQuestionEncoder, "CLIPModel": CLIPModel,
               "ECAEncoder": ECAEncoder, "IntermediateLinearFusion": IntermediateLinearFusion}


def __getattr__(name):
    """``viquae_amd.encoders.ArcFaceR50`` (the face encoder lives in viquae_amd/arcface.py, which imports this module)."""
    if name == "ArcFaceR50":
        from .arcface import ArcFaceR50
        return ArcFaceR50
    raise AttributeError(name)
