"""Forward / backward program of the X-LXMERT hot path over the HIP kernels.

The engine owns a static activation plan for one (B, L, V) batch geometry and runs the encoder
(HF:498-557), the model shell (HF:691-822), the codebook head (ref lxrt/modeling.py:38-53) and the
masked-visual-token losses (ref :237-290) as explicit kernel sequences -- forward and hand-derived backward --
through an `ops` object (xlxmert_amd.ops.HipOps).  No autograd graph, no per-op allocation.

Layout decisions (MI355X-first):
  * language rows (B*L) and visual rows (B*V) of the cross-modality layers live in ONE [B*L + B*V, d] buffer:
    the bidirectional cross-attention uses one set of weights for both directions (HF:386-397), so its QKV
    projection, output projection and LayerNorm run as single contractions over all rows, and the weight
    gradient is one contraction instead of two accumulated ones.
  * query/key/value weights are adjacent in the flat parameter buffer -> one [3d, d] operand.
  * the language side of the last cross layer is skipped when only the visual output is consumed
    (vis_mask task) -- its parameters sit outside the optimizer range.
"""
import math
import os

import torch

from .ops import EPI_DGELU, EPI_GELU, EPI_GELU_DG, EPI_MULAUX, EPI_NONE, EPI_RESIDUAL, EPI_ROWMAX, EPI_TANH, GemmCall


class _Att:
    """LxmertAttention + LxmertAttentionOutput parameters (fused q/k/v views)."""

    def __init__(self, eng, prefix, self_name):
        st = eng.store
        qkv_w = [f"{prefix}.{self_name}.{n}.weight" for n in ("query", "key", "value")]
        qkv_b = [f"{prefix}.{self_name}.{n}.bias" for n in ("query", "key", "value")]
        self.wqkv, self.gwqkv = st.fused(qkv_w, st.compute), st.fused(qkv_w, st.grad)
        self.bqkv, self.gbqkv = st.fused(qkv_b, st.master), st.fused(qkv_b, st.grad)
        o = f"{prefix}.output"
        self.wo, self.gwo = st.cview(o + ".dense.weight"), st.gview(o + ".dense.weight")
        self.bo, self.gbo = st.view(o + ".dense.bias"), st.gview(o + ".dense.bias")
        self.g, self.gg = st.view(o + ".LayerNorm.weight"), st.gview(o + ".LayerNorm.weight")
        self.b, self.gb = st.view(o + ".LayerNorm.bias"), st.gview(o + ".LayerNorm.bias")


_RESERVED_STREAMS = {}


def reserve_streams(device, comm=False):
    """The engine's three extra HIP streams (language stream + the two weight-gradient companions), created and FIRST USED
    here.  comm=True: a fourth one for the library's own RCCL binding (xl_comm_*, trainer XL_COMM=rccl), reserved in the same breath
    so that the collectives' event waits never sit in a compute stream's hardware queue (needs GPU_MAX_HW_QUEUES >= 5).  HIP binds a stream to one of its few hardware queues (GPU_MAX_HW_QUEUES = 4) at first use, round-robin: with the
    main stream these four must land on four different queues, or two of them serialise behind each other's barrier
    packets and the step loses its overlap (+3.3 ms measured).  Any stream that gets used in between shifts the assignment -
    RCCL's does: call this BEFORE torch.distributed.init_process_group (bench.py does; measured 24.4 -> 20.5 ms per step
    with a process group present).  Engines on one device share the streams."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _RESERVED_STREAMS:
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        for st in streams:
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1.0)
        torch.cuda.synchronize(dev)
        _RESERVED_STREAMS[key] = streams
    if comm and len(_RESERVED_STREAMS[key]) == 3:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            torch.zeros(8, device=dev).add_(1.0)
        torch.cuda.synchronize(dev)
        _RESERVED_STREAMS[key].append(st)
    return _RESERVED_STREAMS[key][:3]


def comm_stream(device):
    """the stream reserved for xl_comm_* collectives (reserve_streams(comm=True)), created now if nobody reserved it"""
    dev = torch.device(device)
    reserve_streams(dev, comm=True)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    return _RESERVED_STREAMS[key][3]


class SelfAttBlock:
    """LxmertSelfAttentionLayer (HF:304-316): y = LN(dense(attn(x,x,mask)) + x).  `masked` = language rows (key padding mask, or
    -- packed language rows -- per-example row offsets instead: engine.Engine pack_lang)."""

    def __init__(self, eng, prefix, n_tok, masked, tag):
        self.e, self.n, self.masked, self.tag = eng, n_tok, masked, tag
        self.p = _Att(eng, prefix, "self")
        Mc = eng.MLc if masked else eng.MV          # row capacity (the active row count of a packed language side varies per batch)
        d = eng.d
        self.qkv = eng.act(Mc, 3 * d)
        self.ctx = eng.act(Mc, d)
        self.z = eng.act(Mc, d)
        self.lse = eng.f32(eng.B * eng.H * n_tok)
        self.keep = eng.keep_bits(n_tok, n_tok)
        self.mean, self.rstd = eng.f32(Mc), eng.f32(Mc)
        self.site = eng.new_site(2)

    @property
    def M(self):
        return self.e.ML if self.masked else self.e.MV

    def _att_args(self):
        """(key mask, packed-row keyword arguments) of this block's attention core"""
        e = self.e
        if not self.masked:
            return e.vkmask, {}
        if e.packed:
            return None, dict(q_off=e.loff, k_off=e.loff, q_pad=e.ML, k_pad=e.ML)
        return e.kmask, {}

    def fwd(self, x, y):
        self.e.run_steps(self.fwd_steps(x, y))

    def fwd_steps(self, x, y):
        """the forward as a generator: every dense contraction is YIELDED (ops.GemmCall) instead of launched, everything else is
        issued as the generator advances -- Engine.run_steps launches each yielded call, Engine.run_pair advances a visual and a
        language block of this class in lock step and launches their contractions two at a time (xl_gemm_pair)."""
        e, p, d, M = self.e, self.p, self.e.d, self.M
        ops, tag = e.ops, self.tag
        qkv, ctx, z = self.qkv[:M], self.ctx[:M], self.z[:M]
        yield GemmCall(x, p.wqkv, qkv, p.bqkv, None, None, M, 3 * d, d, d, d, 3 * d, tag=tag)
        km, vl = self._att_args()
        ops.block = tag
        ops.sdpa_fwd(qkv, qkv[:, d:], qkv[:, 2 * d:], km, ctx, self.lse, e.B, e.H, self.n, self.n,
                     e.dh, 3 * d, 3 * d, 3 * d, d, e.scale, e.p_attn, e.seed(self.site), keep_bits=e.kb(self.keep), **vl)
        yield GemmCall(ctx, p.wo, z, p.bo, x, None, M, d, d, d, d, d, ldr=d, epilogue=EPI_RESIDUAL,
                       p_drop=e.p_hid, seed=e.seed(self.site + 1), tag=tag)
        ops.block = tag
        ops.layernorm_fwd(z, p.g, p.b, y, self.mean, self.rstd, M, d, e.eps)
        self.x = x

    def probs(self):
        """attention probabilities [B, H, n, n] fp32 of the last forward (after dropout, as HF returns them)"""
        e, d, M = self.e, self.e.d, self.M
        qkv = self.qkv[:M]
        out = torch.zeros(e.B, e.H, self.n, self.n, dtype=torch.float32, device=e.dev)
        km, vl = self._att_args()
        vl = {k: v for k, v in vl.items() if k in ("q_off", "k_off")}
        e.ops.attn_probs(qkv, qkv[:, d:], km, self.lse, out, e.B, e.H, self.n, self.n, e.dh, 3 * d, 3 * d, e.scale, e.p_attn,
                         e.seed(self.site), **vl)
        return out

    def bwd(self, dy, dx):
        self.e.run_steps(self.bwd_steps(dy, dx))

    def bwd_steps(self, dy, dx):
        e, p, d, M = self.e, self.p, self.e.d, self.M
        ops, tag = e.ops, self.tag
        ops.block = tag
        e.wgrad_sync()                  # the previous block's weight-gradient GEMMs still read the shared scratch
        qkv, ctx, z = self.qkv[:M], self.ctx[:M], self.z[:M]
        dz = e.tmp("dz", M, d)
        dzm = e.ln_bwd_dense(dy, z, p.g, self.mean, self.rstd, dz, p.gg, p.gb, p.gbo, M, d, self.site + 1)
        e.wgrad_defer(dzm, ctx, p.gwo, d, d, M, d, d, d)
        dctx = e.tmp("dctx", M, d)
        yield GemmCall(dzm, p.wo, dctx, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0, tag=tag)
        dqkv = e.tmp("dqkv", M, 3 * d)
        km, vl = self._att_args()
        ops.block = tag
        ops.sdpa_bwd(qkv, qkv[:, d:], qkv[:, 2 * d:], km, dctx, self.lse, dqkv, dqkv[:, d:],
                     dqkv[:, 2 * d:], e.B, e.H, self.n, self.n, e.dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d,
                     e.scale, e.p_attn, e.seed(self.site), bias_grad=p.gbqkv, ws=e.ws, keep_bits=e.kb(self.keep), **vl)      # + d(b_q | b_k | b_v)
        e.wgrad_defer(dqkv, self.x, p.gwqkv, 3 * d, d, M, 3 * d, d, d)
        e.wgrad_flush(pair=True)        # this layer's four weight gradients: launched together with the next layer's
        yield GemmCall(dqkv, p.wqkv, dx, None, dz, None, M, d, 3 * d, 3 * d, d, d, ldr=d, a_kmajor=1, b_kmajor=0,
                       epilogue=EPI_RESIDUAL, tag=tag)


class FFNBlock:
    """LxmertIntermediate + LxmertOutput (HF:325-342): y = LN(W2 gelu(W1 x + b1) + b2 + x).  `lang`: language rows (their
    active count follows the batch when the language rows are packed)."""

    def __init__(self, eng, p_inter, p_out, lang, tag):
        self.e, self.lang, self.tag = eng, lang, tag
        st = eng.store
        self.w1, self.gw1 = st.cview(p_inter + ".dense.weight"), st.gview(p_inter + ".dense.weight")
        self.b1, self.gb1 = st.view(p_inter + ".dense.bias"), st.gview(p_inter + ".dense.bias")
        self.w2, self.gw2 = st.cview(p_out + ".dense.weight"), st.gview(p_out + ".dense.weight")
        self.b2, self.gb2 = st.view(p_out + ".dense.bias"), st.gview(p_out + ".dense.bias")
        self.g, self.gg = st.view(p_out + ".LayerNorm.weight"), st.gview(p_out + ".LayerNorm.weight")
        self.b, self.gb = st.view(p_out + ".LayerNorm.bias"), st.gview(p_out + ".LayerNorm.bias")
        d, dff = eng.d, eng.dff
        Mc = eng.MLc if lang else eng.MV
        self.pre = eng.act(Mc, dff)
        self.h = eng.act(Mc, dff)
        self.z = eng.act(Mc, d)
        self.mean, self.rstd = eng.f32(Mc), eng.f32(Mc)
        self.site = eng.new_site(1)
        self.rows = None                # row count of a forward on a row SUBSET (Engine: the last cross layer's visual side
                                        # on the masked rows only); None = all rows of the side

    @property
    def M(self):
        if self.rows is not None:
            return self.rows
        return self.e.ML if self.lang else self.e.MV

    def fwd(self, x, y):
        self.e.run_steps(self.fwd_steps(x, y))

    def fwd_steps(self, x, y):
        """(a generator of the block's dense contractions: see SelfAttBlock.fwd_steps)"""
        e, d, dff, M = self.e, self.e.d, self.e.dff, self.M
        ops, tag = e.ops, self.tag
        pre, h, z = self.pre[:M], self.h[:M], self.z[:M]
        # self.pre holds gelu'(pre-activation): erf and exp(-x^2/2) are in registers in the forward epilogue anyway, and the
        # backward epilogue becomes a multiply (no second erf + exp per element of the [M, dff] gradient)
        yield GemmCall(x, self.w1, h, self.b1, None, pre, M, dff, d, d, d, dff, ldx=dff, epilogue=EPI_GELU_DG, tag=tag)
        yield GemmCall(h, self.w2, z, self.b2, x, None, M, d, dff, dff, dff, d, ldr=d, epilogue=EPI_RESIDUAL,
                       p_drop=e.p_hid, seed=e.seed(self.site), tag=tag)
        ops.block = tag
        ops.layernorm_fwd(z, self.g, self.b, y, self.mean, self.rstd, M, d, e.eps)
        self.x = x

    def bwd(self, dy, dx):
        self.e.run_steps(self.bwd_steps(dy, dx))

    def bwd_steps(self, dy, dx):
        e, d, dff, M = self.e, self.e.d, self.e.dff, self.M
        ops, tag = e.ops, self.tag
        ops.block = tag
        e.wgrad_sync()
        pre, h, z = self.pre[:M], self.h[:M], self.z[:M]
        # own scratch names: the two weight gradients registered here are launched together with the attention block's
        # (which runs next on this stream and flushes), so dz / dzm / dpre must outlive that block's scratch use
        dz = e.tmp("f_dz", M, d)
        dzm = e.ln_bwd_dense(dy, z, self.g, self.mean, self.rstd, dz, self.gg, self.gb, self.gb2, M, d, self.site,
                             tmp_name="f_dzm")
        e.wgrad_defer(dzm, h, self.gw2, d, dff, M, d, dff, dff)
        dpre = e.tmp("dpre", M, dff)
        yield GemmCall(dzm, self.w2, dpre, None, None, pre, M, dff, d, d, dff, dff, ldx=dff, a_kmajor=1, b_kmajor=0,
                       epilogue=EPI_MULAUX, colsum=self.gb1, ws=e.ws_gemm, tag=tag)      # d(b1) = column sums of dpre, in the same epilogue
        e.wgrad_defer(dpre, self.x, self.gw1, dff, d, M, dff, d, d)
        yield GemmCall(dpre, self.w1, dx, None, dz, None, M, d, dff, dff, d, d, ldr=d, a_kmajor=1, b_kmajor=0,
                       epilogue=EPI_RESIDUAL, tag=tag)


class CrossAttBlock:
    """LxmertXLayer.cross_att (HF:377-398): ONE LxmertCrossAttentionLayer applied in both directions on the
    pre-update inputs.  Rows = [visual (B*V) ; language (ML)] -- the visual rows first, so that the language rows, whose count
    follows the batch when they are packed (Engine pack_lang), are the tail of every buffer."""

    def __init__(self, eng, prefix, need_lang, tag, need_vis=True):
        """need_lang / need_vis: whether the language / visual rows of the OUTPUT are consumed.  The last cross layer of a
        masked-visual-token step needs only the visual side, the one of the pooled-output / language tasks (VQA, word_mask,
        matched) only the language side: the other direction of the attention, its output projection and LayerNorm are
        skipped (their parameters get no gradient in the reference either)."""
        assert need_lang or need_vis
        self.e, self.need_lang, self.need_vis, self.tag = eng, need_lang, need_vis, tag
        self.p = _Att(eng, prefix, "att")
        d = eng.d
        self.qkv = eng.act(eng.MXc, 3 * d)
        self.ctx = eng.act(eng.MXc, d)
        self.z = eng.act(eng.MXc, d)
        self.lse_l = eng.f32(eng.B * eng.H * eng.L)
        self.lse_v = eng.f32(eng.B * eng.H * eng.V)
        self.keep_l, self.keep_v = eng.keep_bits(eng.L, eng.V), eng.keep_bits(eng.V, eng.L)
        self.mean, self.rstd = eng.f32(eng.MXc), eng.f32(eng.MXc)
        self.site = eng.new_site(3)

    def _lq(self):
        """packed-row arguments of the attention with LANGUAGE queries over visual keys"""
        e = self.e
        return dict(q_off=e.loff, q_pad=e.ML) if e.packed else {}

    def _lk(self):
        """(key mask, packed-row arguments) of the attention with visual queries over LANGUAGE keys"""
        e = self.e
        return (None, dict(k_off=e.loff, k_pad=e.ML)) if e.packed else (e.kmask, {})

    def fwd(self, X, Y):
        e, p, d = self.e, self.p, self.e.d
        ops, ML, MV, MX = e.ops, e.ML, e.MV, e.MX
        ops.block = self.tag
        L_, V_ = e.lr, e.vr
        qkv_l, qkv_v = L_(self.qkv), V_(self.qkv)
        if not self.need_vis:
            ops.gemm(L_(X), p.wqkv, qkv_l, p.bqkv, None, None, ML, d, d, d, d, 3 * d)                    # Q of language rows
            ops.gemm(V_(X), p.wqkv[d:], qkv_v[:, d:], p.bqkv[d:], None, None, MV, 2 * d, d, d, d, 3 * d)  # K,V of visual rows
            ops.sdpa_fwd(qkv_l, qkv_v[:, d:], qkv_v[:, 2 * d:], e.vkmask, L_(self.ctx), self.lse_l, e.B, e.H, e.L, e.V, e.dh,
                         3 * d, 3 * d, 3 * d, d, e.scale, e.p_attn, e.seed(self.site), keep_bits=e.kb(self.keep_l), **self._lq())
            ops.gemm(L_(self.ctx), p.wo, L_(self.z), p.bo, L_(X), None, ML, d, d, d, d, d, ldr=d, epilogue=EPI_RESIDUAL,
                     p_drop=e.p_hid, seed=e.seed(self.site + 2))
            ops.layernorm_fwd(L_(self.z), p.g, p.b, L_(Y), L_(self.mean), L_(self.rstd), ML, d, e.eps)
            self.X = X
            return
        if self.need_lang:
            ops.gemm(X[:MX], p.wqkv, self.qkv[:MX], p.bqkv, None, None, MX, 3 * d, d, d, d, 3 * d)
            # language queries over visual keys/values (no mask: visual_attention_mask is None in every caller)
            ops.sdpa_fwd(qkv_l, qkv_v[:, d:], qkv_v[:, 2 * d:], e.vkmask, L_(self.ctx), self.lse_l, e.B, e.H, e.L, e.V, e.dh,
                         3 * d, 3 * d, 3 * d, d, e.scale, e.p_attn, e.seed(self.site), keep_bits=e.kb(self.keep_l), **self._lq())
        else:
            ops.gemm(V_(X), p.wqkv, qkv_v, p.bqkv, None, None, MV, d, d, d, d, 3 * d)                    # Q of visual rows
            ops.gemm(L_(X), p.wqkv[d:], qkv_l[:, d:], p.bqkv[d:], None, None, ML, 2 * d, d, d, d, 3 * d)  # K,V of language rows
        # visual queries over language keys/values, padded language keys excluded
        km, vl = self._lk()
        ops.sdpa_fwd(qkv_v, qkv_l[:, d:], qkv_l[:, 2 * d:], km, V_(self.ctx), self.lse_v, e.B, e.H, e.V, e.L, e.dh,
                     3 * d, 3 * d, 3 * d, d, e.scale, e.p_attn, e.seed(self.site + 1), keep_bits=e.kb(self.keep_v), **vl)
        M = MX if self.need_lang else MV           # rows [0, M): visual rows, then (both directions) the language rows
        ops.gemm(self.ctx[:M], p.wo, self.z[:M], p.bo, X[:M], None, M, d, d, d, d, d, ldr=d, epilogue=EPI_RESIDUAL,
                 p_drop=e.p_hid, seed=e.seed(self.site + 2))
        ops.layernorm_fwd(self.z[:M], p.g, p.b, Y[:M], self.mean[:M], self.rstd[:M], M, d, e.eps)
        self.X = X

    def probs(self):
        """cross-attention probabilities of the LANGUAGE queries over the visual keys, [B, H, L, V] fp32 -- what
        LxmertXLayer.forward hands out as its attention output (HF:417-449: `attention_probs = lang_att_output[1:]`)"""
        e, d = self.e, self.e.d
        assert self.need_lang, "this cross layer skipped its language direction (dead-branch elimination)"
        qkv_l, qkv_v = e.lr(self.qkv), e.vr(self.qkv)
        out = torch.zeros(e.B, e.H, e.L, e.V, dtype=torch.float32, device=e.dev)
        vl = {k: v for k, v in self._lq().items() if k == "q_off"}
        e.ops.attn_probs(qkv_l, qkv_v[:, d:], e.vkmask, self.lse_l, out, e.B, e.H, e.L, e.V, e.dh, 3 * d, 3 * d, e.scale, e.p_attn,
                         e.seed(self.site), **vl)
        return out

    def bwd(self, dY, dX):
        e, p, d = self.e, self.p, self.e.d
        ops, ML, MV, MX = e.ops, e.ML, e.MV, e.MX
        ops.block = self.tag
        e.wgrad_sync()
        if not self.need_vis:
            return self._bwd_lang_only(dY, dX)
        L_, V_ = e.lr, e.vr
        M = MX if self.need_lang else MV
        dz_full = e.tmp("dz", MX, d)
        dz = dz_full[:M]
        dzm = e.ln_bwd_dense(dY[:M], self.z[:M], p.g, self.mean[:M], self.rstd[:M], dz, p.gg, p.gb, p.gbo, M, d,
                             self.site + 2)
        e.wgrad_defer(dzm, self.ctx[:M], p.gwo, d, d, M, d, d, d)
        dctx_full = e.tmp("dctx", MX, d)
        dctx = dctx_full[:M]
        ops.gemm(dzm, p.wo, dctx, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
        dqkv = e.tmp("dqkv", MX, 3 * d)
        dqkv_l, dqkv_v = L_(dqkv), V_(dqkv)
        qkv_l, qkv_v = L_(self.qkv), V_(self.qkv)
        km, vl = self._lk()
        ops.sdpa_bwd(qkv_v, qkv_l[:, d:], qkv_l[:, 2 * d:], km, V_(dctx_full), self.lse_v, dqkv_v, dqkv_l[:, d:],
                     dqkv_l[:, 2 * d:], e.B, e.H, e.V, e.L, e.dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, e.scale,
                     e.p_attn, e.seed(self.site + 1), bias_grad=p.gbqkv, ws=e.ws, keep_bits=e.kb(self.keep_v), **vl)
        X = self.X
        if self.need_lang:
            ops.sdpa_bwd(qkv_l, qkv_v[:, d:], qkv_v[:, 2 * d:], e.vkmask, L_(dctx_full), self.lse_l, dqkv_l, dqkv_v[:, d:],
                         dqkv_v[:, 2 * d:], e.B, e.H, e.L, e.V, e.dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, e.scale,
                         e.p_attn, e.seed(self.site), bias_grad=p.gbqkv, ws=e.ws, keep_bits=e.kb(self.keep_l), **self._lq())      # both directions share the projections
            e.wgrad_defer(dqkv, X[:MX], p.gwqkv, 3 * d, d, MX, 3 * d, d, d)
            e.wgrad_flush()
            ops.gemm(dqkv, p.wqkv, dX[:MX], None, dz_full, None, MX, d, 3 * d, 3 * d, d, d, ldr=d, a_kmajor=1, b_kmajor=0,
                     epilogue=EPI_RESIDUAL)
        else:
            e.wgrad_defer(dqkv_v, V_(X), p.gwqkv, d, d, MV, 3 * d, d, d)
            e.wgrad_defer(dqkv_l[:, d:], L_(X), p.gwqkv[d:], 2 * d, d, ML, 3 * d, d, d)
            e.wgrad_flush()
            ops.gemm(dqkv_v, p.wqkv, V_(dX), None, dz, None, MV, d, d, 3 * d, d, d, ldr=d, a_kmajor=1, b_kmajor=0,
                     epilogue=EPI_RESIDUAL)
            ops.gemm(dqkv_l[:, d:], p.wqkv[d:], L_(dX), None, None, None, ML, d, 2 * d, 3 * d, d, d, a_kmajor=1, b_kmajor=0)

    def _bwd_lang_only(self, dY, dX):
        e, p, d = self.e, self.p, self.e.d
        ops, ML, MV, MX = e.ops, e.ML, e.MV, e.MX
        L_, V_ = e.lr, e.vr
        X = self.X
        dz = L_(e.tmp("dz", MX, d))
        dzm = e.ln_bwd_dense(L_(dY), L_(self.z), p.g, L_(self.mean), L_(self.rstd), dz, p.gg, p.gb, p.gbo, ML, d, self.site + 2)
        e.wgrad_defer(dzm, L_(self.ctx), p.gwo, d, d, ML, d, d, d)
        dctx = L_(e.tmp("dctx", MX, d))
        ops.gemm(dzm, p.wo, dctx, None, None, None, ML, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
        dqkv = e.tmp("dqkv", MX, 3 * d)
        dqkv_l, dqkv_v = L_(dqkv), V_(dqkv)
        qkv_l, qkv_v = L_(self.qkv), V_(self.qkv)
        ops.sdpa_bwd(qkv_l, qkv_v[:, d:], qkv_v[:, 2 * d:], e.vkmask, dctx, self.lse_l, dqkv_l, dqkv_v[:, d:], dqkv_v[:, 2 * d:],
                     e.B, e.H, e.L, e.V, e.dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, e.scale, e.p_attn, e.seed(self.site),
                     bias_grad=p.gbqkv, ws=e.ws, keep_bits=e.kb(self.keep_l), **self._lq())
        e.wgrad_defer(dqkv_l, L_(X), p.gwqkv, d, d, ML, 3 * d, d, d)
        e.wgrad_defer(dqkv_v[:, d:], V_(X), p.gwqkv[d:], 2 * d, d, MV, 3 * d, d, d)
        e.wgrad_flush()
        ops.gemm(dqkv_l, p.wqkv, L_(dX), None, dz, None, ML, d, d, 3 * d, d, d, ldr=d, a_kmajor=1, b_kmajor=0, epilogue=EPI_RESIDUAL)
        ops.gemm(dqkv_v[:, d:], p.wqkv[d:], V_(dX), None, None, None, MV, d, 2 * d, 3 * d, d, d, a_kmajor=1, b_kmajor=0)


class AnswerHead:
    """LxmertVisualAnswerHead on pooled_output (HF:602-614: Linear(d,2d) -> GeLU -> LayerNorm(2d) -> Linear(2d,A)) with
    BCEWithLogitsLoss on soft targets -- the VQA/GQA fine-tune step (ref tasks/vqa_model.py:22-72, vqa.py:166-187)."""

    def __init__(self, eng, num_answers, pair=False):
        self.e, self.A, self.pair = eng, num_answers, pair
        st, d = eng.store, eng.d
        # pair head (NLVR2, ref tasks/nlvr2_model.py:80-86): pooled_output [2P, d] of the flattened (statement, image) rows is
        # read as [P, 2d] -- a view of the same contiguous buffer -- by a head whose first Linear is (2d, 2d)
        assert not pair or eng.B % 2 == 0, "the pair head needs an even number of encoder rows (two images per statement)"
        B = self.Bh = eng.B // 2 if pair else eng.B
        self.din = 2 * d if pair else d
        p = "answer_head.logit_fc"
        self.w0, self.gw0 = st.cview(p + ".0.weight"), st.gview(p + ".0.weight")
        self.b0, self.gb0 = st.view(p + ".0.bias"), st.gview(p + ".0.bias")
        self.g, self.gg = st.view(p + ".2.weight"), st.gview(p + ".2.weight")
        self.b, self.gb = st.view(p + ".2.bias"), st.gview(p + ".2.bias")
        self.w3, self.gw3 = st.cview(p + ".3.weight"), st.gview(p + ".3.weight")
        self.b3, self.gb3 = st.view(p + ".3.bias"), st.gview(p + ".3.bias")
        self.Ap = (num_answers + 7) // 8 * 8
        self.pre, self.h, self.hn = eng.act(B, 2 * d), eng.act(B, 2 * d), eng.act(B, 2 * d)
        self.mean, self.rstd = eng.f32(B), eng.f32(B)
        self.logit = torch.zeros(B, num_answers, dtype=torch.float32, device=eng.dev)
        self.targets = torch.zeros(B, num_answers, dtype=torch.float32, device=eng.dev)
        self.dlogit = eng.act(B, self.Ap)
        self.dhn, self.dh, self.dpre = eng.act(B, 2 * d), eng.act(B, 2 * d), eng.act(B, 2 * d)
        self.dpooled, self.dz = eng.act(eng.B, d), eng.act(eng.B, d)
        self.loss = eng.f32(1)
        # pretraining QA branch (ref lxrt/modeling.py:292-304): CrossEntropyLoss over one answer id per example (-100 = none)
        self.labels = torch.full((B,), -100, dtype=torch.int64, device=eng.dev)
        self.counts, self._ones, self._nm = eng.f32(4), torch.ones(B, dtype=torch.uint8, device=eng.dev), eng.f32(B)
        self.row_argmax = torch.zeros(B, dtype=torch.int32, device=eng.dev)
        self.row_lse, self.row_maxprob = eng.f32(B), eng.f32(B)

    def fwd(self, pooled):
        e, d, B, A, din = self.e, self.e.d, self.Bh, self.A, self.din
        ops = e.ops
        pooled = pooled.view(B, din)
        ops.gemm(pooled, self.w0, self.h, self.b0, None, self.pre, B, 2 * d, din, din, din, 2 * d, ldx=2 * d, epilogue=EPI_GELU)
        ops.layernorm_fwd(self.h, self.g, self.b, self.hn, self.mean, self.rstd, B, 2 * d, 1e-12)
        ops.gemm(self.hn, self.w3, self.logit, self.b3, None, None, B, A, 2 * d, 2 * d, 2 * d, A, out_f32=True)
        return self.logit

    def loss_fwd_bwd(self, want_grad=True):
        B, A = self.Bh, self.A
        self.e.ops.zero(self.loss)
        self.e.ops.bce_logits_fwd_bwd(self.logit, self.targets, self.dlogit if want_grad else None, self.loss, B, A, A, A, self.Ap)
        return self.loss

    def ce_loss_fwd_bwd(self, want_grad=True):
        """qa_loss = CrossEntropyLoss()(answer_score, qa_labels) (ref lxrt/modeling.py:295-298; ignore_index -100) and its
        d(logit); also records qa_pred = argmax (ref :300)."""
        B, A = self.Bh, self.A
        ops = self.e.ops
        ops.zero(self.loss)
        ops.mask_counts(self.labels, self._ones, self.counts, self._nm, B, 1)          # counts[0] = #labels != -100
        if want_grad and self.Ap > A:
            ops.zero(self.dlogit)                                                      # pad columns
        ops.ce_fwd_bwd(self.logit, self.labels, self.counts, self.dlogit if want_grad else None, self.loss, self.row_lse,
                       self.row_argmax, self.row_maxprob, B, A, A, self.Ap, 1.0)
        return self.loss

    def bwd(self, pooled, cls_rows, d_cls, accumulate=False):
        """consumes dlogit; accumulates the head's and the pooler's parameter gradients; writes (or, accumulate=True, adds)
        d(lang_output[:, 0]) into d_cls (a [B, d] view with row stride L*d)."""
        self.bwd_to_pooled(pooled)
        self.e.pooler_backward(self.dpooled, self.dz, cls_rows, d_cls, accumulate=accumulate)

    def bwd_to_pooled(self, pooled):
        """consumes dlogit; accumulates the head's parameter gradients; leaves d(pooled_output) in self.dpooled."""
        e, d, B, A, Ap, din = self.e, self.e.d, self.Bh, self.A, self.Ap, self.din
        ops, st = e.ops, e.store
        pooled, dpooled = pooled.view(B, din), self.dpooled.view(B, din)
        ops.colsum(self.dlogit, self.gb3_pad(), B, Ap, Ap, ws=e.ws_wide("answer", Ap))      # pad columns of dlogit are zero
        e.wgrad_defer(self.dlogit, self.hn, self.gw3, A, 2 * d, B, Ap, 2 * d, 2 * d)
        ops.gemm(self.dlogit, self.w3, self.dhn, None, None, None, B, 2 * d, A, Ap, 2 * d, 2 * d, a_kmajor=1, b_kmajor=0)
        ops.layernorm_bwd(self.dhn, self.h, self.g, self.mean, self.rstd, self.dh, self.gg, self.gb, None, B, 2 * d, ws=e.ws)
        ops.gelu_bwd(self.dh, self.pre, self.dpre, B * 2 * d)
        ops.colsum(self.dpre, self.gb0, B, 2 * d, 2 * d, ws=e.ws)
        e.wgrad_defer(self.dpre, pooled, self.gw0, 2 * d, din, B, 2 * d, din, din)
        ops.gemm(self.dpre, self.w0, dpooled, None, None, None, B, din, 2 * d, 2 * d, din, din, a_kmajor=1, b_kmajor=0)

    def gb3_pad(self):
        """bias-gradient view padded to the 8-column granule of dlogit (the bias unit is padded in the flat buffer)."""
        m = self.e.store.index["answer_head.logit_fc.3.bias"]
        return self.e.store.grad[m.offset:m.offset + self.Ap]


class LangHeads:
    """LxmertPreTrainingHeads (HF:589-657) for the `word_mask` and `matched` pretraining branches (ref lxrt/modeling.py:
    211-235): MLM = decoder(LN(gelu(dense(lang)))) + bias with the decoder TIED to the word-embedding matrix, vocab-way CE
    over the masked tokens; matched = seq_relationship(pooled_output), 2-way CE."""

    def __init__(self, eng):
        self.e = eng
        st, d, ML, B = eng.store, eng.d, eng.MLd, eng.B
        self.has_mlm = "cls.predictions.bias" in st.index and eng.task in ("word_mask", "all")
        self.has_rel = "cls.seq_relationship.weight" in st.index and eng.task in ("matched", "all")
        self.loss = eng.f32(2)                  # [lm_loss, matched_loss]
        self.counts, self.dummy, self.rel_counts = eng.f32(4), eng.f32(B), eng.f32(4)
        if self.has_mlm:
            t = "cls.predictions.transform"
            self.wt, self.gwt = st.cview(t + ".dense.weight"), st.gview(t + ".dense.weight")
            self.bt, self.gbt = st.view(t + ".dense.bias"), st.gview(t + ".dense.bias")
            self.g, self.gg = st.view(t + ".LayerNorm.weight"), st.gview(t + ".LayerNorm.weight")
            self.b, self.gb = st.view(t + ".LayerNorm.bias"), st.gview(t + ".LayerNorm.bias")
            self.vb = st.view("cls.predictions.bias")
            self.Vn = eng.cfg.vocab_size
            self.Vp = (self.Vn + 7) // 8 * 8
            self.pre, self.h, self.hn = eng.act(ML, d), eng.act(ML, d), eng.act(ML, d)
            self.mean, self.rstd = eng.f32(ML), eng.f32(ML)
            self.scores = torch.zeros(ML, self.Vp, dtype=torch.float32, device=eng.dev)       # row stride padded to 8
            self.dscores = eng.act(ML, self.Vp)
            self.word_labels = torch.full((B, eng.L), -100, dtype=torch.int64, device=eng.dev)
            # masked-row mode (training step, row list from the data loader): the 30522-way decoder, its loss and their
            # backward run on the labelled rows only - exact, the loss reads nothing else (same idea as Engine._hrows)
            self.rows = torch.zeros(ML, dtype=torch.int32, device=eng.dev)
            self.labels_c = torch.zeros(ML, dtype=torch.int64, device=eng.dev)
            self.n_rows = 0
        if self.has_rel:
            self.wr, self.gwr = st.cview("cls.seq_relationship.weight"), st.gview("cls.seq_relationship.weight")
            self.br, self.gbr = st.view("cls.seq_relationship.bias"), st.gview("cls.seq_relationship.bias")
            self.rel = torch.zeros(B, 8, dtype=torch.float32, device=eng.dev)                # 2 columns used
            self.drel = eng.act(B, 8)
            self.matched_labels = torch.zeros(B, dtype=torch.int64, device=eng.dev)
            self.dpooled, self.dz = eng.act(B, d), eng.act(B, d)

    def _gvb_pad(self):
        m = self.e.store.index["cls.predictions.bias"]
        return self.e.store.grad[m.offset:m.offset + self.Vp]

    # ---- word_mask
    def set_rows(self, word_rows):
        """word_rows: flat indices b*L+l of the labelled positions (host tensor / list: its length is the launch size), or None."""
        self.n_rows = 0
        if word_rows is not None and 0 < len(word_rows) < self.e.MLd:
            idx = torch.as_tensor(word_rows, dtype=torch.int64)
            n = int(idx.numel())
            self.n_rows = self.e.pad_rows(self.rows, idx, n, self.e.MLd)

    def mlm_fwd(self, lang):
        e, d = self.e, self.e.d
        ops, st = e.ops, e.store
        M = self.n_rows if self.n_rows else e.MLd
        if self.n_rows:
            self.x = e.tmp("mlm_x", e.MLd, d)[:M]
            ops.gather_rows(lang, self.rows, self.x, M, d, d, d)
        else:
            self.x = lang
        ops.gemm(self.x, self.wt, self.h, self.bt, None, self.pre, M, d, d, d, d, d, ldx=d, epilogue=EPI_GELU)
        ops.layernorm_fwd(self.h, self.g, self.b, self.hn, self.mean, self.rstd, M, d, 1e-12)
        ops.gemm(self.hn, st.cview("bert.embeddings.word_embeddings.weight"), self.scores, self.vb, None, None,
                 M, self.Vn, d, d, d, self.Vp, out_f32=True)
        return self.scores

    def mlm_loss_bwd(self, d_lang):
        """CE over the masked tokens (labels -100 ignored) + backward down to d(language_output) (written to d_lang, which
        the caller has zeroed)."""
        self.mlm_loss()
        return self.mlm_bwd(d_lang)

    def mlm_loss(self):
        """lm_loss and d(scores) (nothing of the parameter gradients is touched)."""
        e, Vn, Vp = self.e, self.Vn, self.Vp
        ops = e.ops
        M = self.n_rows if self.n_rows else e.MLd
        ops.zero(self.loss[0:1])
        ops.mask_counts(self.word_labels, e.kmask, self.counts, self.dummy, e.B, e.L)
        labels = self.word_labels
        if self.n_rows:
            labels = self.labels_c[:M]
            ops.gather_labels(self.word_labels, self.rows, labels, M)
        ops.ce_fwd_bwd(self.scores, labels, self.counts, self.dscores, self.loss[0:], None, None, None,
                       M, Vn, Vp, Vp, 1.0)
        return self.loss

    def mlm_bwd(self, d_lang):
        e, d, Vn, Vp = self.e, self.e.d, self.Vn, self.Vp
        ops, st = e.ops, e.store
        M = self.n_rows if self.n_rows else e.MLd
        emb = "bert.embeddings.word_embeddings.weight"
        ops.colsum(self.dscores, self._gvb_pad(), M, Vp, Vp, ws=e.ws_wide("mlm", Vp))
        e.wgrad_defer(self.dscores, self.hn, st.gview(emb), Vn, d, M, Vp, d, d, once=False)  # tied decoder: d(word embeddings) also
                                                                                            # receives the embedding scatter-add
        dhn = e.tmp("dctx", e.MLd, d)
        ops.gemm(self.dscores, st.cview(emb), dhn, None, None, None, M, d, Vn, Vp, d, d, a_kmajor=1, b_kmajor=0)
        dh = e.tmp("dz", e.MLd, d)
        ops.layernorm_bwd(dhn, self.h, self.g, self.mean, self.rstd, dh, self.gg, self.gb, None, M, d, ws=e.ws)
        dpre = e.tmp("dzm", e.MLd, d)
        ops.gelu_bwd(dh, self.pre, dpre, M * d)
        ops.colsum(dpre, self.gbt, M, d, d, ws=e.ws)
        e.wgrad_defer(dpre, self.x, self.gwt, d, d, M, d, d, d)
        e.wgrad_flush()
        if self.n_rows:
            dx = e.tmp("mlm_dx", e.MLd, d)[:M]
            ops.gemm(dpre, self.wt, dx, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
            ops.scatter_rows(dx, self.rows, d_lang, M, d, d, d)
        else:
            ops.gemm(dpre, self.wt, d_lang, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
        return self.loss

    # ---- matched
    def rel_fwd(self, pooled):
        e, d, B = self.e, self.e.d, self.e.B
        e.ops.gemm(pooled, self.wr, self.rel, self.br, None, None, B, 2, d, d, d, 8, out_f32=True)
        return self.rel

    def rel_loss_bwd(self, pooled, cls_rows, d_cls, extra_dpooled=None):
        self.rel_loss()
        return self.rel_bwd(pooled, cls_rows, d_cls, extra_dpooled)

    def rel_loss(self):
        e, B = self.e, self.e.B
        e.ops.zero(self.loss[1:2])
        self.rel_counts.fill_(float(B))                                 # every example carries a matched label
        e.ops.ce_fwd_bwd(self.rel, self.matched_labels, self.rel_counts, self.drel, self.loss[1:], None, None, None, B, 2, 8, 8, 1.0)
        return self.loss

    def rel_bwd(self, pooled, cls_rows, d_cls, extra_dpooled=None):
        e, d, B = self.e, self.e.d, self.e.B
        ops = e.ops
        # the two output rows of seq_relationship: tiny contractions, done on the 8-column padded gradient
        g8 = torch.zeros(8, d, dtype=torch.float32, device=e.dev) if not hasattr(self, "_g8") else self._g8
        self._g8 = g8
        g8.zero_()
        ops.gemm(self.drel, pooled, g8, None, None, None, 8, d, B, 8, d, d, a_kmajor=0, b_kmajor=0, out_f32=True)
        self.gwr.add_(g8[:2])
        gb8 = torch.zeros(8, dtype=torch.float32, device=e.dev) if not hasattr(self, "_gb8") else self._gb8
        self._gb8 = gb8
        gb8.zero_()
        ops.colsum(self.drel, gb8, B, 8, 8, ws=e.ws)
        e.flush_reductions()                     # consumed right away
        self.gbr.add_(gb8[:2])
        w8 = torch.zeros(8, d, dtype=e.cdtype, device=e.dev) if not hasattr(self, "_w8") else self._w8
        self._w8 = w8
        w8[:2].copy_(self.wr)
        ops.gemm(self.drel, w8, self.dpooled, None, None, None, B, d, 8, 8, d, d, a_kmajor=1, b_kmajor=0)
        if extra_dpooled is not None:            # task_qa model: the answer head's d(pooled_output) joins here
            self.dpooled.add_(extra_dpooled)
        e.pooler_backward(self.dpooled, self.dz, cls_rows, d_cls)
        return self.loss


class Engine:
    """Static-shape forward/backward program.  `need_lang`: whether lang/pooled outputs of the last cross layer are
    consumed (False for the masked-visual-token step)."""

    def __init__(self, cfg, store, ops, B, L, V, need_lang=False, train_dropout=False, two_streams=True, pack_lang=None):
        """pack_lang: run the language side on the REAL tokens only.  The attention mask (input_ids > 0, ref lxmert_pretrain.py:206)
        removes the [PAD] positions as keys everywhere (HF:238-266) and no loss or head reads a [PAD] position's output, so every
        row the reference computes for them -- a third of the B x 20 language rows at sentence lengths U{6..20} -- is dead work:
        after the embeddings the language rows are gathered into a packed [sum of lengths, d] matrix (row list + per-example
        offsets from the data loader, rounded up to the GEMM row tile with zero rows), every language-side contraction /
        LayerNorm runs over those rows, the attention cores address an example's rows through the offsets (xl_sdpa_* q_rowoff /
        k_rowoff), and the final language output is scattered back to [B, L, d] with zero rows at the [PAD] positions (the
        reference leaves don't-care values there).  Exact for the real rows and for every gradient.  Default: env XL_PACK_LANG
        (on).  The nn.Module surface runs packed too: hidden_states() scatters the per-layer language states back to the dense
        [B*L, d] layout on request."""
        assert cfg.l_layers >= 1 and cfg.r_layers >= 1 and cfg.x_layers >= 1
        # (text longer than 64 tokens -- ref param.py:140 --max_text_length -- runs on the plain long-sequence attention kernels)
        assert L <= 512 and V <= 64, "text length <= 512 (position table), visual grid <= 64 tokens (samplers / on-chip attention)"
        self.cfg, self.store, self.ops = cfg, store, ops
        self.dev, self.cdtype = store.device, store.compute_dtype
        self.B, self.L, self.V = B, L, V
        self.d, self.dff, self.F, self.K = cfg.hidden_size, cfg.intermediate_size, cfg.visual_feat_dim, cfg.num_clusters
        self.H, self.dh = cfg.num_attention_heads, cfg.head_dim
        self.P = cfg.visual_pos_dim
        self.pack_lang = (os.environ.get("XL_PACK_LANG", "1") != "0") if pack_lang is None else bool(pack_lang)
        # attention dropout decisions saved by the forward for the backward (Engine.keep_bits); 0: the backward hashes again (A/B)
        self.use_keep_bits = os.environ.get("XL_SDPA_KEEP_BITS", "1") != "0"
        # XL_PAIR_BLOCKS=1 (opt-in): visual / language sub-blocks of one shape class in lock step on one stream, their contractions
        # two per launch (run_pair).  Measured (round 5, bs 256, profiles/r05a): the paired launches take 1.6 ms less GEMM time per
        # step (14.7 -> 13.1 ms isolated, roofline.frac 0.30 -> 0.34) and the step gets SLOWER -- 16.75 -> 17.6 ms with the language
        # blocks' own kernels (LayerNorm, attention core) queued behind the visual ones on the chain, 18.2 ms with them on the
        # language stream (XL_PAIR_SIDE=1: ~170 cross-stream hand-overs per step at 5-9 us each); with those kernels REMOVED
        # altogether (tools/sensitivity.py no_lang_small_paired: the bound for launches that merge them too) 16.68 ms, i.e. no
        # better than the language stream of rounds 1-4 beside the visual chain.  Default: off.
        self.pair_blocks = os.environ.get("XL_PAIR_BLOCKS", "0") != "0"
        self.pair_side = os.environ.get("XL_PAIR_SIDE", "0") != "0"       # ... the language block's own kernels on the language stream
        self.packed = False                 # this batch runs packed (set_inputs: pack_lang and a usable attention mask)
        self.MLd, self.MV = B * L, B * V    # dense language rows / visual rows
        # language row CAPACITY of every buffer (a packed row count is rounded up to the row tile) and the ACTIVE count
        self.MLc = (self.MLd + self.ROW_PAD - 1) // self.ROW_PAD * self.ROW_PAD if self.pack_lang else self.MLd
        self.ML = self.MLd
        self.MXc = self.MV + self.MLc
        self.scale = 1.0 / math.sqrt(self.dh)
        self.eps = cfg.layer_norm_eps
        self.need_lang = need_lang
        self.p_hid = cfg.hidden_dropout_prob if train_dropout else 0.0
        self.p_attn = cfg.attention_probs_dropout_prob if train_dropout else 0.0
        # optional callback(lo, hi, flush, lane): gradients in flat range [lo, hi) are final in the order of the CURRENT stream;
        # flush = nothing more will be appended to this range.  Two growing ranges: main stream [0, language_range.lo) and
        # then [language_range.hi, n_used); language stream language_range itself (exchange stream, see _ready_lang)
        self.grad_ready = None
        self._lane_lo = None
        self.grad_is_zero = False       # set by the trainer when its optimizer pass cleared the gradient buffer
        self._n_sites = 2               # sites 0 / 1: embedding and visual-feature-encoder output dropout
        self._seed = 0
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)       # step part of the dropout seeds
        self._tmp = {}
        self.overwritten = set()          # flat gradient ranges some backward has STORED (overwrite mode): never cleared again
        self.written_now = set()          # ... and the ones the backward(s) since the last optimizer pass stored
        self._pending = {"v": [], "l": []}
        self._held, self._held_layers, self._gen = {"v": [], "l": []}, {"v": 0, "l": 0}, {"v": 0, "l": 0}
        self._pending_block = {"v": "", "l": ""}
        self.act_bytes = 0
        st, d = store, self.d
        # ---- blocks
        e_ = "bert.encoder"
        self.lang_layers = [(SelfAttBlock(self, f"{e_}.layer.{i}.attention", L, True, f"l{i}"),
                             FFNBlock(self, f"{e_}.layer.{i}.intermediate", f"{e_}.layer.{i}.output", True, f"l{i}"))
                            for i in range(cfg.l_layers)]
        self.vis_layers = [(SelfAttBlock(self, f"{e_}.r_layers.{i}.attention", V, False, f"r{i}"),
                            FFNBlock(self, f"{e_}.r_layers.{i}.intermediate", f"{e_}.r_layers.{i}.output", False, f"r{i}"))
                           for i in range(cfg.r_layers)]
        self.x_layers = []
        task0 = getattr(store, "task", "vis_mask")
        for i in range(cfg.x_layers):
            p = f"{e_}.x_layers.{i}"
            lang_on = need_lang or i < cfg.x_layers - 1
            # VQA / word_mask / matched read only the language (pooled) output: the visual side of the last cross layer is dead
            vis_on = not (task0 in ("vqa", "word_mask", "matched", "qa") and i == cfg.x_layers - 1)
            blk = {"cross": CrossAttBlock(self, p + ".visual_attention", lang_on, f"x{i}", need_vis=vis_on), "lang_on": lang_on,
                   "vis_on": vis_on}
            if vis_on:
                blk["sa_v"] = SelfAttBlock(self, p + ".visn_self_att", V, False, f"x{i}v")
                blk["ffn_v"] = FFNBlock(self, p + ".visn_inter", p + ".visn_output", False, f"x{i}v")
            if lang_on:
                blk["sa_l"] = SelfAttBlock(self, p + ".lang_self_att", L, True, f"x{i}l")
                blk["ffn_l"] = FFNBlock(self, p + ".lang_inter", p + ".lang_output", True, f"x{i}l")
            self.x_layers.append(blk)
        # ---- activations of the chain
        self.emb_y, self.emb_pre = self.act(self.MLd, d), self.act(self.MLd, d)        # embeddings: dense [B*L] rows
        self.emb_mean, self.emb_rstd = self.f32(self.MLd), self.f32(self.MLd)
        self.emb_p = self.act(self.MLc, d) if self.pack_lang else None                 # ... gathered to the packed rows
        self.lrows = torch.full((self.MLc,), -1, dtype=torch.int32, device=self.dev)   # b*L+l of packed row r (-1: pad tail)
        self.loff = torch.zeros(B + 1, dtype=torch.int32, device=self.dev)             # first packed row of example b
        self.lang_pad = self.act(self.MLd, d) if self.pack_lang else None              # final language output, dense layout
        self.glang_pad = self.act(self.MLd, d) if self.pack_lang else None             # its gradient, dense layout
        self.feats = self.act(self.MV, self.F)
        self.xv = self.act(self.MV, d)
        self.vis0 = self.act(self.MV, d)
        self.vn_stats = [self.f32(self.MV) for _ in range(4)]
        self.lang_mid = [self.act(self.MLc, d) for _ in range(cfg.l_layers)]          # attention-block outputs
        self.lang_out = [self.act(self.MLc, d) for _ in range(cfg.l_layers - 1)]
        self.vis_mid = [self.act(self.MV, d) for _ in range(cfg.r_layers)]
        self.vis_out = [self.act(self.MV, d) for _ in range(cfg.r_layers - 1)]
        self.X = [self.act(self.MXc, d) for _ in range(cfg.x_layers + 1)]            # [vis ; lang] per cross layer
        self.XY = [self.act(self.MXc, d) for _ in range(cfg.x_layers)]               # cross-attention outputs
        self.XS = [self.act(self.MXc, d) for _ in range(cfg.x_layers)]               # self-attention outputs
        self.pooled = self.act(B, d)
        self.kmask = torch.ones(B, L, dtype=torch.uint8, device=self.dev)
        self.vkmask, self._vkmask_buf = None, None        # visual_attention_mask (HF:760-770): None in every reference caller
        self.pos = torch.zeros(self.MV, self.P, dtype=torch.float32, device=self.dev)
        self.ids = torch.zeros(B, L, dtype=torch.int64, device=self.dev)
        self.word_order = torch.arange(B * L, dtype=torch.int32, device=self.dev)      # token positions sorted by (id, position)
        self.tt = torch.zeros(B, L, dtype=torch.int64, device=self.dev)
        self.cid = torch.zeros(B, V, dtype=torch.int64, device=self.dev)
        self.vmask = torch.zeros(B, V, dtype=torch.uint8, device=self.dev)
        self.labels = torch.full((B, V), -100, dtype=torch.int64, device=self.dev)
        self._hrows, self._hvis = None, None
        self.feat_tgt, self._feat_tgt_buf = None, None
        self.compact_head = os.environ.get("XL_COMPACT_HEAD", "1") != "0"   # training step: codebook head on the masked rows only
        # ... and with it the visual feed-forward block of the LAST cross layer: nothing but the head reads its output, so on
        # the other rows it is as dead as the head is (encoder_forward(ffn_rows=)); 0: all rows, as the reference computes them
        self.compact_last_ffn = os.environ.get("XL_COMPACT_LAST_FFN", "1") != "0"
        self._ffn_rows_run, self._ffn_in_c, self._vis_c, self._dvis_c = None, None, None, None
        self.task = getattr(store, "task", "vis_mask")
        # answer head on pooled_output: the VQA/GQA fine-tune model, or a pretraining model built with task_qa (then its CE
        # loss joins every task's loss, ref lxrt/modeling.py:292-304)
        self.answer = AnswerHead(self, store.num_answers, getattr(store, "pair", False)) if getattr(store, "num_answers", 0) > 0 else None
        self.task_qa = self.answer is not None and self.task != "vqa"
        self.lang_heads = LangHeads(self) if self.task in ("word_mask", "matched", "all") else None
        if self.task in ("vqa", "word_mask", "matched", "qa", "all") or self.task_qa:
            assert need_lang, "this task reads the language / pooled output: build the engine with need_lang=True"
        # ---- head (ref lxrt/modeling.py:38-53) + losses
        h = "obj_predict_head"
        self.hd = {k: (st.cview(n) if c else st.view(n), st.gview(n)) for k, n, c in (
            ("wt", h + ".transform.dense.weight", True), ("bt", h + ".transform.dense.bias", False),
            ("gt", h + ".transform.LayerNorm.weight", False), ("bbt", h + ".transform.LayerNorm.bias", False),
            ("wf", h + ".linear_feat.weight", True), ("bf", h + ".linear_feat.bias", False),
            ("bc", h + ".out_cluster.bias", False))}
        self.Kp = (self.K + 7) // 8 * 8
        self.t_pre, self.t_h, self.t_y = self.act(self.MV, d), self.act(self.MV, d), self.act(self.MV, d)
        self.t_mean, self.t_rstd = self.f32(self.MV), self.f32(self.MV)
        self.feat = self.act(self.MV, self.F)
        n_head_rows = self.MV if self.task in ("vis_mask", "all") else 8     # the codebook head is only read by vis_mask steps
        self.logits = torch.zeros(n_head_rows, self.K, dtype=torch.float32, device=self.dev)
        self.dlogits = self.act(n_head_rows, self.Kp)
        self.dfeat = self.act(self.MV, self.F)
        self.counts, self.nmask = self.f32(4), self.f32(B)
        self.losses = self.f32(4)             # [obj_loss, feat_loss, -, -]
        self.row_lse, self.row_maxprob = self.f32(self.MV), self.f32(self.MV)
        self.row_argmax = torch.zeros(self.MV, dtype=torch.int32, device=self.dev)
        self.mrows = torch.zeros(self.MV, dtype=torch.int32, device=self.dev)         # ids b*V+v of the masked positions
        self.labels_c = torch.zeros(self.MV, dtype=torch.int64, device=self.dev)
        self.n_mrows = 0
        self.mf_tmp = self.f32(d)
        self.mf_tmp_c = self.act(1, d)
        # ---- activation-gradient ping-pong
        self.GA, self.GB = self.act(self.MXc, d), self.act(self.MXc, d)
        # two-stage column reductions: one workspace per stream (language / visual work runs concurrently)
        # (the second stages are deferred and combined per layer -- flush_reductions -- so every producer between two flushes
        # gets a workspace region of its own: _WS_REGIONS per stream)
        # regions are allocated when first used, per (stream, scratch generation, index): a generation's regions are rewritten only
        # after the launch that combined them is known to be done (wgrad_sync).  The 10k-codebook head's column sum (the only
        # producer wider than dff) has a workspace of its own.
        self._ws_len = {"v": ops.workspace_floats(max(3 * d, self.dff, self.F)), "l": ops.workspace_floats(max(3 * d, self.dff))}
        self._ws, self._ws_wide = {}, None
        self._ws_i = {"v": 0, "l": 0}
        self._red_gens = {"v": set(), "l": set()}      # (generation, producing stream) with column-sum partials not yet combined
        self._gen_guard = {"v": {}, "l": {}}           # generation -> event after the companion-stream launch that read / combined it
        self._guard_ring, self._guard_next = {"v": [], "l": []}, {"v": 0, "l": 0}
        self._deferred = False
        # The language stream (B*20 rows) fills less than half the chip per kernel; its layers are independent of the
        # visual stream inside the L/R stacks and between two cross-attention blocks, so they run on a second HIP stream.
        self._tag = "v"
        # Weight-gradient GEMMs (dW = dY^T X) are off the dX dependency chain: each stream gets a companion stream for
        # them, so they co-run with the chain's next kernels (and their epilogue bursts interleave).
        self.side, self._dw = None, None
        self._dw_busy = {"v": False, "l": False}
        if two_streams and self.dev.type == "cuda":
            self.side, dwv, dwl = reserve_streams(self.dev)
            self._dw = {"v": dwv, "l": dwl}
        # slab workspaces (xl_gemm_set_workspace) of every stream this engine launches contractions on: the tail split of
        # the 10k-codebook contractions, and -- opt-in, XL_GEMM_WGRAD_SLABS=1 -- deterministic weight gradients
        self._slab_ws = []
        if self.dev.type == "cuda" and hasattr(ops, "gemm_workspace") and os.environ.get("XL_GEMM_SLABS", "1") != "0":
            streams = list(self._dw.values()) if self._dw is not None else []
            streams += [torch.cuda.current_stream()] + ([self.side] if self.side is not None else [])
            for s_ in streams:
                self._slab_ws.append(ops.gemm_workspace(256, s_))

    # ------------------------------------------------------------ row layout: [visual (MV) ; language (ML active of MLc)]
    @property
    def MX(self):
        return self.MV + self.ML

    def vr(self, T):
        """visual rows of a [vis ; lang] buffer"""
        return T[:self.MV]

    def lr(self, T):
        """active language rows of a [vis ; lang] buffer"""
        return T[self.MV:self.MV + self.ML]

    # ------------------------------------------------------------ memory helpers
    def act(self, *shape):
        t = torch.zeros(*shape, dtype=self.cdtype, device=self.dev)
        self.act_bytes += t.numel() * t.element_size()
        return t

    def f32(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.dev)

    def keep_bits(self, nq, nk):
        """buffer in which an attention core's forward leaves its dropout decisions for its backward (xl_sdpa_fwd / _bwd keep_bits:
        the backward then tests a bit instead of evaluating the mask hash twice per element), or None where the kernels have no
        use for it (fp32 parity mode, the host restatement of the CPU tests)"""
        sizer = getattr(self.ops, "sdpa_keep_bits_bytes", None)
        n = sizer(self.B, self.H, nq, nk, self.dh) if sizer is not None else 0
        if n == 0:
            return None
        self.act_bytes += n
        return torch.zeros(n // 4, dtype=torch.int32, device=self.dev)

    def kb(self, buf):
        return buf if self.use_keep_bits else None

    _WS_REGIONS = 16

    def _ws_region(self, t, g, i):
        key = (t, g, i)
        if key not in self._ws:
            self._ws[key] = self.f32(self._ws_len[t])
        return self._ws[key]

    @property
    def ws(self):
        """workspace of the next two-stage column reduction on the current stream (a fresh region while deferred)."""
        t = self._tag
        if not self._deferred:
            return self._ws_region(t, 0, 0)
        i = self._ws_i[t]
        assert i < self._WS_REGIONS, "too many column reductions in one scratch generation"
        self._ws_i[t] = i + 1
        self._note_partials(t)
        return self._ws_region(t, self._gen[t], i)

    def _cur_handle(self):
        return torch.cuda.current_stream().cuda_stream if self.dev.type == "cuda" else 0

    # a contraction with fused column sums that run_pair holds back is LAUNCHED on the pair's stream, whatever stream its block's
    # generator was advanced on: the library files its pending second stage under the launching stream
    _gemm_handle = None

    @property
    def ws_gemm(self):
        """ws for the fused column sums of a contraction yielded to run_steps / run_pair"""
        h, self._note_handle = self._note_handle, self._gemm_handle
        try:
            return self.ws
        finally:
            self._note_handle = h

    _note_handle = None

    def _note_partials(self, t):
        """lane t has column-sum partials of its current scratch generation waiting for their combine, filed by the library under
        the stream that launched the producer (a lane's kernels may run on either compute stream: run_pair)"""
        h = self._note_handle if self._note_handle is not None else self._cur_handle()
        self._red_gens[t].add((self._gen[t], h))

    def _combined(self, handle, ev=None):
        """everything pending from the stream `handle` has been combined (ev: by a launch on a companion stream, done when ev is)"""
        for t in ("v", "l"):
            for g, h in [x for x in self._red_gens[t] if x[1] == handle]:
                if ev is not None:
                    self._gen_guard[t].setdefault(g, []).append(ev)
                self._red_gens[t].discard((g, h))

    def ws_wide(self, name, N):
        """workspace of a column sum wider than dff (the 10k-codebook head's, the MLM decoder's, the answer head's): one per user,
        one producer per step each, combined before the next step's producer runs"""
        if self._ws_wide is None:
            self._ws_wide = {}
        if name not in self._ws_wide:
            self._ws_wide[name] = self.f32(self.ops.workspace_floats(N))
        if self._deferred:
            self._note_partials(self._tag)
        return self._ws_wide[name]

    def defer_reductions(self, on):
        """backward of a training step: second stages of the column reductions are combined per layer (one launch)."""
        import os
        if on and os.environ.get("XL_DEFER_REDUCE", "1") == "0":
            return
        if on != self._deferred:
            if not on:
                self.flush_reductions()
            self._deferred = on
            self.ops.set_deferred_reduce(1 if on else 0)

    def flush_reductions(self):
        """combine everything pending from the current stream, on the current stream (in order behind the producers)"""
        if self._deferred:
            self.ops.flush_reductions()
            self._combined(self._cur_handle())      # (the library combines everything pending from this stream, whichever lane's)

    def _flush_if_reporting(self):
        """per-layer combine on the CURRENT stream -- only when somebody is told that the layer's gradients are final (gradient
        exchange).  Otherwise the pending combines ride on the weight-gradient companion stream with the next grouped launch
        (wgrad_flush), off the dX dependency chain; the end of the backward combines what is left."""
        if self.grad_ready is not None or self._dw is None or self.side is None:
            self.flush_reductions()

    class _LangStream:
        def __init__(self, eng):
            self.e = eng
            self.ctx = torch.cuda.stream(eng.side) if eng.side is not None else None

        def __enter__(self):
            self.e._tag = "l"
            if self.ctx is not None:
                self.ctx.__enter__()

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            self.e._tag = "v"

    def lang_stream(self):
        return Engine._LangStream(self)

    class _Lane:
        """bookkeeping lane ("v" visual / "l" language: backward scratch sets, column-sum workspace regions, pending weight
        gradients and their companion stream) WITHOUT a change of stream: a language-side block of a pair runs on the stream
        of its visual partner but keeps the language lane's buffers and weight-gradient grouping."""

        def __init__(self, eng, tag):
            self.e, self.tag = eng, tag

        def __enter__(self):
            self.prev, self.e._tag = self.e._tag, self.tag

        def __exit__(self, *a):
            self.e._tag = self.prev

    def lane(self, tag):
        return Engine._Lane(self, tag)

    def run_steps(self, steps):
        """a block's generator (SelfAttBlock / FFNBlock fwd_steps, bwd_steps) on its own: every yielded contraction is one launch"""
        ops = self.ops
        for c in steps:
            ops.block = c.tag
            ops.gemm(*c.a, **c.kw)

    def run_pair(self, steps_v, steps_l):
        """a VISUAL-side and a LANGUAGE-side block of one class (the self-attention / FFN sub-blocks of a cross-modality layer,
        HF:417-449, or a layer of each single-modality stack, HF:516-529) in lock step on the current stream: the two generators
        are advanced alternately -- each issues its own LayerNorm / attention-core / bookkeeping calls, the language one inside
        the language lane -- and every pair of contractions they yield goes out as ONE xl_gemm_pair launch: the language side's
        39 row tiles ride in the CUs that the visual side's 64 x (3 | 9 | 12) tiles leave idle in their last round, instead of
        holding 39-156 CUs at a tenth of the matrix rate on a stream of their own."""
        ops = self.ops
        side = self.side if self.pair_side else None
        self._gemm_handle = self._cur_handle()          # (fused column sums of a yielded contraction: filed under THIS stream)
        try:
            while True:
                # The language block's own kernels (LayerNorm, attention core: ~3300 rows, 5-25 us each) go to the language stream,
                # beside the visual block's on this one -- queued one behind the other they added ~100 us per layer to the chain
                # (measured: +0.85 ms per step, more than the paired contractions save).  The language stream continues behind the
                # pair launch just queued; this stream waits for it only if the block issued anything before its next contraction.
                issued = False
                if side is not None:
                    self.fork()
                    with self.lang_stream():
                        n0 = ops.ncalls
                        cl = next(steps_l, None)
                        issued = ops.ncalls != n0
                else:
                    with self.lane("l"):
                        cl = next(steps_l, None)
                cv = next(steps_v, None)
                if issued:
                    self.join()
                if cv is None and cl is None:
                    return
                if cv is not None and cl is not None:
                    ops.block = f"{cv.tag}+{cl.tag}"
                    ops.gemm_pair(cv, cl)
                else:
                    c = cv if cv is not None else cl
                    ops.block = c.tag
                    ops.gemm(*c.a, **c.kw)
        finally:
            self._gemm_handle = None

    def _lang_side(self):
        """context of the language lane's own work in a paired section: the language stream (pair_side) or this stream"""
        return self.lang_stream() if (self.pair_side and self.side is not None) else self.lane("l")

    # Set by a trainer whose optimizer pass runs behind the step on its own stream (trainer.PretrainStep overlap_optimizer): called
    # with the key of a parameter group right before the forward first reads it, on the stream that reads it.
    params_ready = None

    def _pr(self, key):
        if self.params_ready is not None:
            self.params_ready(key)

    def fork(self):
        """language stream waits for everything queued so far on the main (visual) stream."""
        if self.side is not None:
            self.ops.stream_fork(torch.cuda.current_stream(), self.side)

    def join(self):
        """main stream waits for everything queued so far on the language stream."""
        if self.side is not None:
            self.ops.stream_fork(self.side, torch.cuda.current_stream())

    # Set by the trainer for a backward that starts from gradients nobody needs (the first backward after an optimizer pass):
    # every weight gradient with ONE contribution per step is stored instead of accumulated (xl_gemm_wgrad_group overwrite_mask),
    # so its part of the buffer is neither cleared by the optimizer pass nor re-read by the epilogue (4 + 4 bytes per parameter).
    # `overwritten` collects the flat ranges written that way (the trainer marks them "do not clear": ParamStore.mark_overwritten).
    dw_overwrite = False

    def wgrad_defer(self, dY, X, dW, M, N, K, lda, ldb, ldc, once=True):
        """register dW[M,N] += dY[K,M]^T X[K,N]; launched with the block's other weight gradients by wgrad_flush().
        once: this tensor gets no other gradient contribution in the step (False: the MLM decoder tied to the word embeddings)."""
        self._pending[self._tag].append((dY, X, dW, M, N, K, lda, ldb, ldc, bool(once and self.dw_overwrite), self._gen[self._tag]))
        self._pending_block[self._tag] = getattr(self.ops, "block", "")

    def _flat_range(self, dW, M, N, ldc):
        if ldc == N and dW.dtype == torch.float32:
            off = (dW.data_ptr() - self.store.grad.data_ptr()) // 4
            if 0 <= off and off + M * N <= self.store.n_total:
                return (off, off + M * N)
        return None

    def _note_overwritten(self, dW, M, N, ldc):
        r = self._flat_range(dW, M, N, ldc)
        if r is not None:
            self.overwritten.add(r)
            self.written_now.add(r)

    def _is_kept(self, dW, M, N, ldc):
        """this weight gradient's range was stored (not accumulated) by an earlier backward: the optimizer pass no longer clears it"""
        return self._flat_range(dW, M, N, ldc) in self.overwritten

    PAIR_LAYERS = int(os.environ.get("XL_WGRAD_PAIR", "2"))          # layers per weight-gradient launch (1: every layer its own)
    # Backward scratch generations.  The weight-gradient launches read a layer's scratch (dz, dqkv, dpre ...) from the companion
    # stream long after the dX chain has moved on, so the chain writes every layer's scratch into the NEXT of NGEN buffer sets and,
    # before it reuses a set, waits for the launch that read it (an event per launch, _gen_guard) -- NGEN layers later, when that
    # launch is long done.  With two sets (rounds 1-3) the chain waited for every grouped launch at the start of the very next
    # block: 0.37 ms of weight gradients on 216 CUs with nothing else running, then the chain's LayerNorm / attention kernels with
    # the matrix units idle.  8 sets cost ~6 GB of the 288.
    NGEN = max(2, int(os.environ.get("XL_SCRATCH_GENS", "8")))

    def _advance_gen(self, tag):
        g = (self._gen[tag] + 1) % self.NGEN
        if any(pr[10] == g for pr in self._held[tag]):
            # weight gradients HELD for a grouped launch with the next layer still read the scratch set that comes up for reuse
            # (few sets -- XL_SCRATCH_GENS below ~5 -- and a cross-modality layer, whose cross-attention block closes a second
            # set per layer): launch them now, so that the set has a guard event for the next writer to wait on
            self.wgrad_flush(pair=True, force=True)
        stale = [h for gg, h in self._red_gens[tag] if gg == g]
        for h in stale:                         # its column-sum partials were never combined (no launch since): do it now, in order
            if h == self._cur_handle():
                self.flush_reductions()
            else:
                # produced on the OTHER compute stream (a language-side head's backward on the language stream, then this lane's
                # blocks paired onto the visual chain, or the reverse): combined on the producing stream, behind its producers, and
                # this stream -- about to reuse the set's workspace regions -- waits for that combine
                st = torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream()
                ev = self._guard_event(tag)
                with torch.cuda.stream(st):
                    self.ops.flush_reductions()
                self.ops.event_record(ev, st)
                self.ops.stream_wait(ev, torch.cuda.current_stream())
                self._combined(h)
        self._gen[tag] = g
        self._ws_i[tag] = 0

    def _guard_event(self, tag):
        """an event for a companion-stream launch (or a cross-stream combine) of lane `tag`.  Re-recording an event that an older guard
        list still refers to would make wgrad_sync wait on the wrong point (with two events per generation the ring of
        2 * NGEN can wrap onto a live reference when few scratch sets are used): such candidates are passed over, and the ring grows
        when every member is referenced."""
        ring = self._guard_ring[tag]
        if len(ring) < 2 * self.NGEN:
            ring.append(self.ops.new_event())
            return ring[-1]
        live = {e for evs in self._gen_guard[tag].values() for e in evs}          # (events are integer handles)
        for _ in range(len(ring)):
            self._guard_next[tag] += 1
            ev = ring[self._guard_next[tag] % len(ring)]
            if ev not in live:
                return ev
        ring.append(self.ops.new_event())
        return ring[-1]

    def wgrad_flush(self, pair=False, force=False):
        """queue the registered weight gradients as ONE grouped launch on the companion stream of the current stream
        (after everything queued so far): off the dX dependency chain.  An FFN block leaves its two problems pending for the
        attention block of the same layer: four per layer.
        pair=True (self-attention + FFN layers): the layer's four problems are HELD and launched together with the next
        layer's -- 216 output tiles of 256x256 fill the chip without a K split, so every tile has one writer and the launch needs
        no atomics (two launches with a K split of 2 and a pass of fp32 atomics each: 2 x 247 us per visual layer pair, one
        launch: ~390 us).  Every call closes a scratch generation (NGEN above); force=True launches whatever is held (end of a
        stream's backward)."""
        tag = self._tag
        if pair:
            self._held[tag] += self._pending[tag]
            self._pending[tag] = []
            self._held_layers[tag] += 0 if force else 1
            if not self._held[tag] or (self._held_layers[tag] < self.PAIR_LAYERS and not force):
                if not force:
                    self._advance_gen(tag)      # the next layer writes the next scratch set
                return
            probs, self._held[tag], self._held_layers[tag] = self._held[tag], [], 0
        else:
            probs, self._pending[tag] = self._pending[tag], []
        if not probs:
            return
        self.ops.block = self._pending_block[tag]
        dw = self._dw.get(tag) if (self._dw is not None and self.side is not None) else None
        ride = dw is not None and self._deferred and self.grad_ready is None
        for i in range(0, len(probs), 8):       # (xl_gemm_wgrad_group takes up to 8 problems)
            chunk = [pr[:9] for pr in probs[i:i + 8]]
            mask = sum(1 << j for j, pr in enumerate(probs[i:i + 8]) if pr[9])
            if mask and not self.ops.wgrad_group_one_writer(chunk):
                # A K-split launch would have to clear C first (a memset per problem): leave a range that nobody has promised to
                # overwrite to the optimizer pass.  A range ALREADY marked "kept" (ParamStore.mark_overwritten: the optimizer pass
                # does not clear it any more) keeps its bit -- the row count of the packed language side moves K across the
                # K / 512 split threshold from batch to batch, and accumulating into the uncleared buffer would add the previous
                # step's gradient; with the bit set xl_gemm_wgrad_group clears C itself in front of the split launch.
                mask = sum(1 << j for j, pr in enumerate(probs[i:i + 8]) if pr[9] and self._is_kept(pr[2], pr[3], pr[4], pr[8]))
            for j, pr in enumerate(probs[i:i + 8]):
                if (mask >> j) & 1:
                    self._note_overwritten(pr[2], pr[3], pr[4], pr[8])
                else:
                    # accumulating (a later micro-batch of an accumulation window, a tied tensor) into a range the optimizer pass no
                    # longer clears and no backward since that pass has stored: it still holds the previous step's gradient
                    r = self._flat_range(pr[2], pr[3], pr[4], pr[8])
                    if r in self.overwritten and r not in self.written_now:
                        self.ops.zero(self.store.grad[r[0]:r[1]])
                        self.written_now.add(r)
            kw = {"overwrite_mask": mask} if mask else {}
            if dw is None:
                self.ops.gemm_wgrad_group(chunk, **kw)
                continue
            cur = torch.cuda.current_stream()
            self.ops.stream_fork(cur, dw)
            with torch.cuda.stream(dw):
                self.ops.gemm_wgrad_group(chunk, **kw)
                if ride and i + 8 >= len(probs):
                    # the column-sum combines pending on this stream ride behind the weight gradients, off the dX chain
                    self.ops.flush_reductions_on(cur)
            self._dw_busy[tag] = True
        if dw is not None:
            ev = self._guard_event(tag)
            self.ops.event_record(ev, dw)
            for g in {pr[10] for pr in probs}:
                self._gen_guard[tag].setdefault(g, []).append(ev)
            if ride:            # everything pending from the producing stream was combined behind this launch, whichever lane's
                self._combined(self._cur_handle(), ev)
        if not force:
            self._advance_gen(tag)

    def wgrad_sync(self):
        """the current stream is about to write the current scratch generation: wait for the companion-stream launch that read it
        (weight gradients) or combined its column-sum partials, if there was one since the generation was last waited for."""
        tag = self._tag
        for ev in self._gen_guard[tag].pop(self._gen[tag], ()):
            self.ops.stream_wait(ev, torch.cuda.current_stream())

    def wgrad_sync_all(self):
        """current stream waits for EVERYTHING queued so far on its companion stream (gradients final: reports, end of backward)"""
        tag = self._tag
        if self._dw_busy[tag]:
            self.ops.stream_fork(self._dw[tag], torch.cuda.current_stream())
            self._dw_busy[tag] = False
        self._gen_guard[tag].clear()

    def tmp(self, name, M, N):
        """backward scratch, shared by all blocks of one stream (sized for the largest user); NGEN sets per stream, one per scratch
        generation (wgrad_flush)."""
        key = (name, N, self._tag, self._gen[self._tag])
        if key not in self._tmp:
            self._tmp[key] = torch.zeros(self.MXc, N, dtype=self.cdtype, device=self.dev)
        return self._tmp[key][:M]

    ROW_PAD = 256
    _NO_VIS_GRAD = object()             # (_dvis_c) the backward of a row-subset forward without a vision-output gradient

    def pad_rows(self, buf, idx, n, cap):
        """row list of a masked-row head -> device buffer `buf` (int32), its length rounded up to ROW_PAD (the GEMM row tile)
        with -1 entries: the kernels treat those as zero rows / skip them (include/xlxmert_hip.h xl_gather_labels), so the
        launch sizes of a step take few distinct values and a captured step (hipGraph) can be replayed.  Returns the padded
        length (0 for an empty list; `cap` = all rows when the padding would reach it)."""
        if n == 0:
            return 0
        npad = min(cap, (n + self.ROW_PAD - 1) // self.ROW_PAD * self.ROW_PAD)
        buf[:n].copy_(idx, non_blocking=True)
        if npad > n:
            buf[n:npad].fill_(-1)
        return npad

    def new_site(self, n):
        s = self._n_sites
        self._n_sites += n
        return s

    # Dropout seeds: mask(site, row, col) = hash(site_seed + 1000003 * step_seed, row, col).  The site part is a launch
    # argument; the step part lives in DEVICE memory (self.seed_dev) and is read by the kernels, so a captured step (hipGraph)
    # draws fresh masks on every replay.  The library samples the pointer when a launch is issued: it is set for the duration
    # of this engine's forward / backward only (direct users of the ops keep plain seeds).
    def seed(self, site):
        return site * 7919 + 12345

    def set_step_seed(self, seed):
        """step part of the dropout seeds (the trainer passes step * world + rank); a tiny fill on the current stream."""
        self._seed = int(seed) & 0x7FFFFFFFFFFF
        self.seed_dev.fill_(self._seed)

    class _Seeded:
        def __init__(self, eng):
            self.e = eng

        def __enter__(self):
            self.e.ops.set_step_seed_ptr(self.e.seed_dev)

        def __exit__(self, *a):
            self.e.ops.set_step_seed_ptr(None)

    def ln_bwd_dense(self, dy, z, g, mean, rstd, dz, gg, gb, gbias, M, N, site, tmp_name="dzm"):
        """LayerNorm backward of a `LN(dropout(dense(.)) + residual)` block: returns the gradient entering the dense layer
        (dz itself when dropout is off, else the masked copy written by the same kernel); the dense bias gradient is fused."""
        if self.p_hid == 0:
            self.ops.layernorm_bwd(dy, z, g, mean, rstd, dz, gg, gb, gbias, M, N, ws=self.ws)
            return dz
        dzm = self.tmp(tmp_name, M, N)
        self.ops.layernorm_bwd(dy, z, g, mean, rstd, dz, gg, gb, gbias, M, N, ws=self.ws, dx_dropped=dzm,
                               p_drop=self.p_hid, seed=self.seed(site))
        return dzm

    def _ready(self, prefix):
        self._ready_upto(self.store.range_of(prefix)[1])

    def _ready_heads(self):
        """every head that sits on top of the encoder (codebook head, cls.*, answer head, pooler) is done: their gradients
        are the first block of the flat buffer (params._backward_rank 0)."""
        self._ready_upto(self.store.heads_end())

    def _ready_upto(self, hi):
        assert not self._pending["v"] and not self._pending["l"], "weight gradients registered but never flushed"
        if self.grad_ready is None and self._dw is not None and self.side is not None:
            return                  # nobody to report to: no combine, no wait here (see _flush_if_reporting)
        self.flush_reductions()
        self.wgrad_sync_all()
        if self.grad_ready is not None and not self._held["v"] and not self._held["l"]:   # (a held layer's weight gradients are
            self._report("v", hi)                                    #  not final yet: the next report covers its range)

    def _report(self, lane, hi, flush=False):
        lo = self._lane_lo[lane]
        if hi > lo or flush:
            self.grad_ready(lo, max(lo, hi), flush, lane)
            self._lane_lo[lane] = max(lo, hi)

    def _ready_lang(self, hi, flush=False):
        """inside lang_stream(): the language-range gradients below `hi` are final once the language stream and its
        weight-gradient companion stream reach this point: the language stream waits for the companion and issues the
        report itself (a third stream for this would need a fifth hardware queue: see reserve_streams)."""
        if self.grad_ready is None or (self._held["l"] and not flush):
            return
        assert not self._held["l"]
        self.wgrad_sync_all()
        self._report("l", hi, flush)

    def sync_compute_weights(self):
        """refresh the compute-dtype copy of the master parameters (after load_state_dict / init)."""
        st = self.store
        if st.master_partial:
            # sharded exchange with gather="bf16": the compute copy IS the replicated state (the all-gather moves it), the fp32 master
            # matrices are whole on their owner only -- casting master -> compute here would revert every non-owned matrix to its
            # initial value and the replicas would diverge.  PretrainStep.gather_state() makes the master copy whole.
            return
        if st.compute_dtype != torch.float32:
            self.ops.cast_from_f32(st.master, st.compute, st.n_total)

    # ------------------------------------------------------------ inputs
    def set_inputs(self, input_ids, attention_mask=None, token_type_ids=None, visual_pos=None, cluster_ids=None,
                   vis_mask=None, obj_labels=None, visual_feats=None, masked_rows=None, feat_labels=None,
                   visual_attention_mask=None, inputs_embeds=None, lang_rows=None, lang_off=None, word_order=None):
        """masked_rows (optional): ascending ids b*V+v of the masked positions, i.e. vis_mask.flatten().nonzero(), as the data
        loader can compute them on the CPU next to vis_mask itself; given, the step needs no host <-> device round trip.
        feat_labels (optional, [B,V,F]): regression targets of the feature loss (label_dict['feat_labels'], ref
        lxrt/modeling.py:275; the trainer passes the real grid features, lxmert_pretrain.py:177-179); without them the
        feature loss regresses onto the centroid of each position's cluster id.
        lang_rows / lang_off (optional, pack_lang): ascending ids b*L+l of the real tokens, i.e. attention_mask.flatten().nonzero(),
        and the [B+1] prefix sums of the per-example token counts -- computed by the data loader next to the mask (given, the
        step needs no host <-> device round trip; every example must have at least one real token).
        word_order (optional): int32 [B*L], the flat token positions sorted by (input id, position) = a stable argsort of
        input_ids.flatten() (trainer.word_order_of) -- the embedding backward gives every word-table row ONE writer that adds
        its occurrences in that order (deterministic scatter, xl_embed_bwd); sorted on the device when the loader sends none."""
        B, L, V = self.B, self.L, self.V
        # inputs_embeds [B, L, d] instead of input_ids (HF:699,731-744,773 -> HF:191-214: they replace the word-embedding
        # lookup; position / token-type embeddings, LayerNorm and dropout still apply): staged as a per-call "word table" of
        # B*L + 1 rows addressed by ids 1..B*L, so the embedding kernels run unchanged and their scatter-add leaves
        # d(inputs_embeds) in rows 1.. of a table-shaped fp32 buffer (row 0 is the kernels' frozen padding_idx row)
        self.embeds_mode = inputs_embeds is not None
        if self.embeds_mode:
            assert input_ids is None and tuple(inputs_embeds.shape) == (B, L, self.d), (inputs_embeds.shape, (B, L, self.d))
            if getattr(self, "_emb_tab", None) is None:
                self._emb_tab = torch.zeros(B * L + 1, self.d, dtype=self.cdtype, device=self.dev)
                self._emb_grad = torch.zeros(B * L + 1, self.d, dtype=torch.float32, device=self.dev)
                self._emb_ids = torch.arange(1, B * L + 1, dtype=torch.int64, device=self.dev).view(B, L)
                self._emb_order = torch.arange(B * L, dtype=torch.int32, device=self.dev)       # (all ids distinct: already sorted)
            self._emb_tab[1:].copy_(inputs_embeds.reshape(B * L, self.d), non_blocking=True)
        else:
            assert tuple(input_ids.shape) == (B, L), (input_ids.shape, (B, L))
            self.ids.copy_(input_ids, non_blocking=True)
            if word_order is None:
                word_order = torch.sort(input_ids.reshape(-1), stable=True).indices
            elif not getattr(self, "_order_checked", False):
                # the embedding backward gives every word-table row ONE writer -- the block that owns the row's run in this order -- and
                # commits with a plain read-modify-write: an order computed from OTHER ids (before MLM masking, re-tokenisation ...) makes
                # several owners race on a row, silently.  One check (a host round trip) on the first loader-supplied order.
                self._order_checked = True
                o = word_order.reshape(-1).to(device=input_ids.device, dtype=torch.int64)
                srt = input_ids.reshape(-1)[o]
                ok = o.numel() == B * L and bool((srt[1:] >= srt[:-1]).all()) and bool((torch.sort(o).values == torch.arange(B * L, device=o.device)).all())
                if not ok:
                    raise ValueError("set_inputs(word_order=): not a permutation that sorts THIS call's input_ids (trainer.word_order_of "
                                     "must be computed from the ids that are passed in, after any masking)")
            self.word_order.copy_(word_order.reshape(-1), non_blocking=True)
        if attention_mask is None:
            self.kmask.fill_(1)
        else:
            self.kmask.copy_(attention_mask.reshape(B, L) != 0, non_blocking=True)
        self.packed, self.ML = False, self.MLd
        if self.pack_lang and attention_mask is not None:
            if lang_rows is None:                   # no row list from the loader: one device round trip
                m = attention_mask.reshape(B, L) != 0
                lang_rows = m.reshape(-1).nonzero().reshape(-1)
                cnt = m.sum(1)
                assert int(cnt.min().item()) >= 1, "an example without a single real token"
                lang_off = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])
            idx = lang_rows.reshape(-1)
            n = int(idx.numel())
            assert 0 < n <= self.MLd and lang_off is not None and int(lang_off.numel()) == B + 1
            self.ML = self.pad_rows(self.lrows, idx, n, self.MLc)
            self.loff.copy_(lang_off.reshape(-1), non_blocking=True)
            self.packed = True
        if token_type_ids is None:
            self.tt.zero_()
        else:
            self.tt.copy_(token_type_ids, non_blocking=True)
        self.pos.copy_(visual_pos.reshape(self.MV, self.P), non_blocking=True)
        self.vkmask = None
        if visual_attention_mask is not None:       # [B, V], nonzero = attend (HF:760-770): masks the visual KEYS of the visual
            if self._vkmask_buf is None:            # self-attention and of the language -> vision cross-attention
                self._vkmask_buf = torch.ones(B, V, dtype=torch.uint8, device=self.dev)
            self._vkmask_buf.copy_(visual_attention_mask.reshape(B, V) != 0, non_blocking=True)
            self.vkmask = self._vkmask_buf
        self.use_codebook = cluster_ids is not None
        if cluster_ids is not None and self.store.centroids_c is None:
            raise RuntimeError("cluster_ids given but the store has no codebook: set_centroids / set_visual_embedding first "
                               "(ref lxrt/modeling.py:140-151, 185-186)")
        if cluster_ids is not None:
            self.cid.copy_(cluster_ids, non_blocking=True)
            self.has_vmask = vis_mask is not None
            if vis_mask is not None:
                self.vmask.copy_(vis_mask.reshape(B, V) != 0, non_blocking=True)
                if self.compact_head and self.task in ("vis_mask", "all"):
                    # the only host <-> device round trip of a step: how many positions are masked (sizes the head's launches)
                    idx = masked_rows.reshape(-1) if masked_rows is not None else (vis_mask.reshape(-1) != 0).nonzero().reshape(-1)
                    self.n_mrows = self.pad_rows(self.mrows, idx, int(idx.numel()), self.MV)
        else:
            self.feats.copy_(visual_feats.reshape(self.MV, self.F), non_blocking=True)
        if obj_labels is not None:
            self.labels.copy_(obj_labels, non_blocking=True)
        self.feat_tgt = None
        if feat_labels is not None:
            if self._feat_tgt_buf is None:
                self._feat_tgt_buf = self.act(self.MV, self.F)
            self._feat_tgt_buf.copy_(feat_labels.reshape(self.MV, self.F), non_blocking=True)
            self.feat_tgt = self._feat_tgt_buf

    def d_inputs_embeds(self):
        """gradient w.r.t. the inputs_embeds of the last backward ([B, L, d] fp32 view; valid until the next backward)."""
        assert getattr(self, "embeds_mode", False)
        return self._emb_grad[1:].view(self.B, self.L, self.d)

    def hidden_states(self):
        """(language_hidden_states, vision_hidden_states) of the last encoder_forward in HF's order (HF:516-544): the output of
        every language layer then of every cross layer / of every visual layer then of every cross layer -- views of the
        activation plan (valid until the next forward)."""
        cfg, ML = self.cfg, self.ML
        lang = [self.lang_out[i][:ML] if i < cfg.l_layers - 1 else self.lr(self.X[0]) for i in range(cfg.l_layers)]
        vis = [self.vis_out[i] if i < cfg.r_layers - 1 else self.vr(self.X[0]) for i in range(cfg.r_layers)]
        for i, blk in enumerate(self.x_layers):
            if blk["lang_on"]:
                lang.append(self.lr(self.X[i + 1]))
            if blk["vis_on"]:
                vis.append(self.vr(self.X[i + 1]))
        if self.packed:                 # packed language rows -> the dense [B*L, d] layout, zero rows at the [PAD] positions
            lang = [self._dense_lang(t) for t in lang]
        return lang, vis

    def attention_probs(self):
        """(language_attentions, vision_attentions, cross_encoder_attentions) of the last encoder_forward, as HF's LxmertEncoder
        collects them (HF:498-557): one [B, H, L, L] per language layer, one [B, H, V, V] per visual layer, one [B, H, L, V]
        (language queries over visual keys) per cross layer.  Recomputed from the saved q / k / log-sum-exp: the fused attention
        kernels never store them, and nothing on the training path asks for them."""
        with Engine._Seeded(self):
            lang = [sa.probs() for sa, _ in self.lang_layers]
            vis = [sa.probs() for sa, _ in self.vis_layers]
            cross = [blk["cross"].probs() for blk in self.x_layers]
        return lang, vis, cross

    def _dense_lang(self, t, out=None):
        """packed language rows -> dense [B*L, d] (zero rows where the attention mask is 0)"""
        if out is None:
            out = torch.zeros(self.MLd, self.d, dtype=self.cdtype, device=self.dev)
        else:
            self.ops.zero(out)
        self.ops.scatter_rows(t, self.lrows, out, self.ML, self.d, self.d, self.d)
        return out

    # ------------------------------------------------------------ forward
    def encoder_forward(self, want_pooled=True, ffn_rows=None):
        """ffn_rows = (rows int32 [n] ascending, padded with -1; n): the caller reads the vision output at these rows only (the
        masked-visual-token step: the codebook head and both of its losses, ref lxrt/modeling.py:253-256, 273-287), so the last
        cross layer's visual feed-forward block -- LxmertIntermediate / LxmertOutput (HF:325-342) act on every row by itself --
        runs on those rows only and leaves its output COMPACT in self._vis_c[:n]; the vision rows of the returned tensors
        are then not written.  Exact: the rows left out feed nothing, forward or backward."""
        with Engine._Seeded(self):
            return self._encoder_forward(want_pooled, ffn_rows)

    def _stack_pairs(self):
        """layers of the two single-modality stacks that run as pairs (run_pair): the LAST min(l_layers, r_layers) of each --
        visual layer r - n + k with language layer l - n + k -- so that the longer stack's first layers (9 language against 5
        visual: layers 0-3) run ahead on the other stream, beside the feature encoder / the embeddings and the previous step's
        optimizer pass, and in the backward pass behind the pairs, beside the feature encoder's backward."""
        if not self.pair_blocks:
            return []
        n = min(self.cfg.l_layers, self.cfg.r_layers)
        return [(self.cfg.r_layers - n + k, self.cfg.l_layers - n + k) for k in range(n)]

    def _lang_layer_io(self, i):
        """(input, attention-block output, output) of language layer i (layer 0 reads the embeddings: _language_stack_forward)"""
        ML = self.ML
        x = (self.emb_p[:ML] if self.packed else self.emb_y) if i == 0 else self.lang_out[i - 1][:ML]
        y = self.lr(self.X[0]) if i == self.cfg.l_layers - 1 else self.lang_out[i][:ML]
        return x, self.lang_mid[i][:ML], y

    def _vis_layer_io(self, i):
        x = self.vis0 if i == 0 else self.vis_out[i - 1]
        y = self.vr(self.X[0]) if i == self.cfg.r_layers - 1 else self.vis_out[i]
        return x, self.vis_mid[i], y

    def _language_stack_forward(self, n_layers=None):
        """embeddings + the first n_layers (default: all l_layers) self-attention layers (HF:516-521); the last layer leaves the
        language rows of the first cross layer's input in the language rows of X[0]."""
        cfg, st, ops, d, ML = self.cfg, self.store, self.ops, self.d, self.ML
        n_layers = cfg.l_layers if n_layers is None else n_layers
        e = "bert.embeddings"
        self._pr("emb")
        emb = getattr(self, "embeds_mode", False)
        ops.embed_ln_fwd(self._emb_ids if emb else self.ids, self.tt,
                         self._emb_tab if emb else st.cview(e + ".word_embeddings.weight"), st.cview(e + ".position_embeddings.weight"),
                         st.cview(e + ".token_type_embeddings.weight"), st.view(e + ".LayerNorm.weight"),
                         st.view(e + ".LayerNorm.bias"), self.emb_y, self.emb_pre, self.emb_mean, self.emb_rstd,
                         self.B, self.L, d, self.eps)
        if self.p_hid > 0:              # HF:213
            ops.dropout(self.emb_y, self.emb_y, self.MLd, d, d, d, self.p_hid, self.seed(0))
        x = self.emb_y
        if self.packed:                 # the real tokens' rows, packed (zero rows in the tail that pads to the row tile)
            x = self.emb_p[:ML]
            ops.gather_rows(self.emb_y, self.lrows, x, ML, d, d, d)
        for i, (sa, ffn) in enumerate(self.lang_layers[:n_layers]):
            self._pr(("lang", i))
            x, mid, y = self._lang_layer_io(i)
            sa.fwd(x, mid)
            ffn.fwd(mid, y)

    def _encoder_forward(self, want_pooled=True, ffn_rows=None):
        cfg, st, ops, d = self.cfg, self.store, self.ops, self.d
        ML, MV = self.ML, self.MV
        X0 = self.X[0]
        last = self.x_layers[-1]
        if ffn_rows is not None and not (self.compact_last_ffn and last["vis_on"] and not (self.pair_blocks and last["lang_on"])):
            ffn_rows = None
        self._ffn_rows_run = ffn_rows
        for blk in self.x_layers:
            if blk["vis_on"]:
                blk["ffn_v"].rows = None
        self.fork()
        # samplers: the text does not change between refinement steps and, without dropout, neither does the output of the
        # language stack (embeddings + l_layers self-attention layers: it never sees the visual tokens) -- it still sits in
        # the language rows of X[0] from the loop's first pass (no later layer writes there), so the later passes skip it.  Exact.
        skip_lang = getattr(self, "_reuse_lang_stack", False) and self.p_hid == 0 and self.p_attn == 0
        pairs = [] if skip_lang else self._stack_pairs()
        n_lang_alone = cfg.l_layers - len(pairs)
        n_vis_alone = cfg.r_layers - len(pairs)
        if not skip_lang:
            with self.lang_stream():            # ---- language stack (HF:516-521) on the side stream: all of it, or the layers
                self._language_stack_forward(n_lang_alone)      # that have no visual partner (_stack_pairs)
        # ---- visual feature encoder + relational stack (HF:513, 524-529) on the main stream
        self._pr("visn")
        if self.use_codebook:
            ops.codebook_gather(self.cid, self.vmask if self.has_vmask else None, st.centroids_c, st.view("mask_feat"),
                                self.feats, MV, self.F)
        v = "bert.encoder.visn_fc"
        ops.block = "visn_fc"
        ops.gemm(self.feats, st.cview(v + ".visn_fc.weight"), self.xv, st.view(v + ".visn_fc.bias"), None, None,
                 MV, d, self.F, self.F, self.F, d)
        ops.visn_ln_fwd(self.xv, self.pos, st.view(v + ".box_fc.weight"), st.view(v + ".box_fc.bias"),
                        st.view(v + ".visn_layer_norm.weight"), st.view(v + ".visn_layer_norm.bias"),
                        st.view(v + ".box_layer_norm.weight"), st.view(v + ".box_layer_norm.bias"),
                        self.vis0, *self.vn_stats, MV, d, self.P, self.eps)
        if self.p_hid > 0:              # HF:475
            ops.dropout(self.vis0, self.vis0, MV, d, d, d, self.p_hid, self.seed(1))
        for i, (sa, ffn) in enumerate(self.vis_layers[:n_vis_alone]):
            self._pr(("vis", i))
            x, mid, y = self._vis_layer_io(i)
            sa.fwd(x, mid)
            ffn.fwd(mid, y)
        self.join()
        for iv, il in pairs:                # ---- a visual and a language layer per step, their contractions two per launch
            self._pr(("vis", iv))
            self._pr(("lang", il))
            (sa_v, ffn_v), (sa_l, ffn_l) = self.vis_layers[iv], self.lang_layers[il]
            xv, mv, yv = self._vis_layer_io(iv)
            xl, ml, yl = self._lang_layer_io(il)
            self.run_pair(sa_v.fwd_steps(xv, mv), sa_l.fwd_steps(xl, ml))
            self.run_pair(ffn_v.fwd_steps(mv, yv), ffn_l.fwd_steps(ml, yl))
        for i, blk in enumerate(self.x_layers):
            Xi, Y, S, Xo = self.X[i], self.XY[i], self.XS[i], self.X[i + 1]
            self._pr(("x", i))              # (the language stream forks from this stream after the wait and inherits it)
            blk["cross"].fwd(Xi, Y)
            if self.pair_blocks and blk["lang_on"] and blk["vis_on"]:
                self.run_pair(blk["sa_v"].fwd_steps(self.vr(Y), self.vr(S)), blk["sa_l"].fwd_steps(self.lr(Y), self.lr(S)))
                self.run_pair(blk["ffn_v"].fwd_steps(self.vr(S), self.vr(Xo)), blk["ffn_l"].fwd_steps(self.lr(S), self.lr(Xo)))
                continue
            if blk["lang_on"]:
                self.fork()
                with self.lang_stream():
                    blk["sa_l"].fwd(self.lr(Y), self.lr(S))
                    blk["ffn_l"].fwd(self.lr(S), self.lr(Xo))
            if blk["vis_on"]:
                blk["sa_v"].fwd(self.vr(Y), self.vr(S))
                if blk is last and ffn_rows is not None:
                    rows, n = ffn_rows
                    if self._ffn_in_c is None:
                        self._ffn_in_c, self._vis_c = self.act(MV, d), self.act(MV, d)
                    ops.block = blk["ffn_v"].tag
                    ops.gather_rows(self.vr(S), rows, self._ffn_in_c[:n], n, d, d, d)       # (pad entries: zero rows)
                    blk["ffn_v"].rows = n
                    blk["ffn_v"].fwd(self._ffn_in_c[:n], self._vis_c[:n])
                else:
                    blk["ffn_v"].fwd(self.vr(S), self.vr(Xo))
            if blk["lang_on"]:
                self.join()
        Xl = self.X[-1]
        self.lang_final, self.vis_final = self.lr(Xl), self.vr(Xl)
        if self.packed and self.need_lang:          # language_output in the reference's [B, L, d] layout for whoever reads it
            self.lang_final = self._dense_lang(self.lr(Xl), self.lang_pad)
        self._pr("heads")                   # pooler and every head on top of the encoder
        if want_pooled and self.need_lang:
            # LxmertPooler (HF:566-572): tanh(dense(lang[:, 0]))
            ops.gemm(self.lang_final, st.cview("bert.pooler.dense.weight"), self.pooled, st.view("bert.pooler.dense.bias"),
                     None, None, self.B, d, d, self.L * d, d, d, epilogue=EPI_TANH)
        return self.lang_final, self.vis_final, self.pooled

    # The head runs either on all B*V visual rows (inference, sampler, the nn.Module API) or, inside the training step, on the
    # masked rows only: both losses read nothing else (ref lxrt/modeling.py:253-256: labels -100 elsewhere; :273-287: the
    # SmoothL1 term is multiplied by vis_mask), so with `--vis_mask_predict` masks (n ~ U{1..64}) half of the head's rows --
    # and of its 10k-codebook contractions -- are never needed.  `_hrows` = None (all rows) or (rows int32, n).
    def _head_rows(self):
        return self._hrows[1] if self._hrows is not None else self.MV

    def head_forward(self, want_logits=True):
        """LxmertVisualObjHead.forward (ref lxrt/modeling.py:38-53): returns (feat, logits)."""
        ops, d, F, K = self.ops, self.d, self.F, self.K
        ops.block = "head"
        M = self._head_rows()
        hd = self.hd
        vis = self.vis_final
        if self._hrows is not None and self._ffn_rows_run is not None:
            assert self._ffn_rows_run[1] == M
            self._hvis_buf = self._vis_c                       # the last visual feed-forward block ran on these rows only
            vis = self._hvis_buf[:M]
        elif self._hrows is not None:
            self._hvis_buf = self.tmp("vis_c", self.MV, d)     # kept: the backward contracts over it (tmp() there may hand out the
            vis = self._hvis_buf[:M]                           # other scratch set -- the generation flips with the layers' backward)
            ops.gather_rows(self.vis_final, self._hrows[0], vis, M, d, d, d)
        self._hvis = vis
        ops.gemm(vis, hd["wt"][0], self.t_h, hd["bt"][0], None, self.t_pre, M, d, d, d, d, d, ldx=d, epilogue=EPI_GELU)
        ops.layernorm_fwd(self.t_h, hd["gt"][0], hd["bbt"][0], self.t_y, self.t_mean, self.t_rstd, M, d, self.eps)
        ops.gemm(self.t_y, hd["wf"][0], self.feat, hd["bf"][0], None, None, M, F, d, d, d, F)
        if want_logits:
            ops.gemm(self.feat, self.store.centroids_c, self.logits, hd["bc"][0], None, None, M, K, F, F, F, K, out_f32=True)
        return self.feat, self.logits

    def losses_forward_backward(self, want_grad=True, feat_loss=True):
        """ref lxrt/modeling.py:237-290.  Leaves d(logits) / d(feat) for head_backward; returns the loss buffer
        [obj_loss, feat_loss] (device, fp32)."""
        ops, F, K = self.ops, self.F, self.K
        M = self._head_rows()
        ops.zero(self.losses)
        ops.mask_counts(self.labels, self.vmask, self.counts, self.nmask, self.B, self.V)
        labels = self.labels
        rows = None
        if self._hrows is not None:
            rows = self._hrows[0]
            labels = self.labels_c[:M]
            ops.gather_labels(self.labels, rows, labels, M)
        ops.ce_fwd_bwd(self.logits, labels, self.counts, self.dlogits if want_grad else None, self.losses[0:],
                       None, None, None, M, K, K, self.Kp, 1.0)
        self.with_feat_loss = feat_loss
        if feat_loss:
            ops.featloss_fwd_bwd(self.feat, self.store.centroids_c, self.cid, self.vmask, self.nmask,
                                 self.dfeat if want_grad else None, self.losses[1:], self.B, self.V, F, 1.0,
                                 rows=rows, n_rows=M if rows is not None else 0, targets=self.feat_tgt)
        return self.losses

    def pooler_backward(self, dpooled, dz, cls_rows, d_cls, accumulate=False):
        """LxmertPooler backward (HF:566-572): pooled = tanh(W_p cls + b_p).  cls_rows / d_cls are [B, d] views of the
        language output / its gradient with row stride L*d (the [CLS] rows); accumulate: d_cls += instead of =."""
        ops, st, d, B, L = self.ops, self.store, self.d, self.B, self.L
        ops.tanh_bwd(dpooled, self.pooled, dz, B * d)
        ops.colsum(dz, st.gview("bert.pooler.dense.bias"), B, d, d, ws=self.ws)
        self.wgrad_defer(dz, cls_rows, st.gview("bert.pooler.dense.weight"), d, d, B, d, L * d, d)
        self.wgrad_flush()
        if accumulate:              # in-place: every output element is read (residual) and written by the same lane
            ops.gemm(dz, st.cview("bert.pooler.dense.weight"), d_cls, None, d_cls, None, B, d, d, d, d, L * d, ldr=L * d,
                     a_kmajor=1, b_kmajor=0, epilogue=EPI_RESIDUAL)
        else:
            ops.gemm(dz, st.cview("bert.pooler.dense.weight"), d_cls, None, None, None, B, d, d, d, d, L * d,
                     a_kmajor=1, b_kmajor=0)

    # ---- QA branch of a task_qa pretraining model (ref lxrt/modeling.py:292-304): rides on every task
    def _qa_forward(self, qa_labels):
        """answer_score = answer_head(pooled_output) and qa_loss / d(answer_score); True when the branch is active."""
        if not self.task_qa:
            assert qa_labels is None, "qa_labels given but the model has no QA head (build the store with num_answers > 0)"
            return False
        assert qa_labels is not None, "a task_qa model adds qa_loss in every branch: label_dict['qa_labels'] is required"
        self.answer.labels.copy_(qa_labels.reshape(-1), non_blocking=True)
        self.answer.fwd(self.pooled)
        self.answer.ce_loss_fwd_bwd(True)
        return True

    def backward_from_outputs(self, d_lang=None, d_vis=None, d_pooled=None):
        """backward of LxmertModel.forward from gradients of its three outputs (any may be None): ACCUMULATES into store.grad
        (HF:691-822; the pooler's gradient joins the [CLS] rows of d(language_output))."""
        assert self.need_lang, "needs the language side of the last cross layer (engine built with need_lang=True)"
        d = self.d
        if self._ffn_rows_run is not None:          # the last forward was a masked-visual-token step's: the vision output exists
            if d_vis is not None:                   # at the masked rows only, in compact form (encoder_forward ffn_rows)
                raise RuntimeError("backward_from_outputs(d_vis=): the last forward computed the vision output at the masked rows "
                                   "only; run encoder_forward() for a backward from a vision-output gradient")
            self._dvis_c = Engine._NO_VIS_GRAD
        self.begin_backward()
        GA = self.GA
        if d_lang is None or d_vis is None:
            self.zero_out_grads(GA)
        if d_lang is not None:
            self.glang(GA).copy_(d_lang.reshape(self.MLd, d))
        if d_vis is not None:
            self.vr(GA).copy_(d_vis.reshape(self.MV, d))
        if d_pooled is not None:
            if not hasattr(self, "_dpooled"):
                self._dpooled, self._dpool_z = self.act(self.B, d), self.act(self.B, d)
            self._dpooled.copy_(d_pooled.reshape(self.B, d))
            cls_rows, d_cls = self._cls_views(GA)
            self.pooler_backward(self._dpooled, self._dpool_z, cls_rows, d_cls, accumulate=True)
        self.encoder_backward(have_lang_grad=True)

    def glang(self, G):
        """where the heads leave d(language_output) in the reference's dense [B*L, d] layout: the language rows of the gradient
        buffer G themselves, or -- packed language rows -- a dense side buffer that encoder_backward gathers into them."""
        return self.glang_pad if self.packed else self.lr(G)

    def zero_out_grads(self, G):
        """d(language_output) = d(vision_output) = 0 (the caller then writes what its loss produces)"""
        self.ops.zero(G[:self.MX])
        if self.packed:
            self.ops.zero(self.glang_pad)

    def _cls_views(self, G):
        """[B, d] views with row stride L*d of the [CLS] rows of language_output and of its gradient"""
        B, L, d = self.B, self.L, self.d
        return self.lang_final.view(B, L * d)[:, :d], self.glang(G).view(B, L * d)[:, :d]

    # Every branch of XLxmertForPretraining.forward (ref lxrt/modeling.py:154-308) as two phases: task_forward = encoder + heads
    # + losses (and the loss gradients w.r.t. the head outputs: no parameter gradient is touched), task_backward = everything
    # that accumulates into store.grad.  The trainer runs them back to back around one clear of the gradient buffer
    # (*_forward_backward below); the nn.Module surface runs them from autograd's forward / backward.
    def task_forward(self, task, word_labels=None, word_rows=None, matched_labels=None, qa_labels=None, feat_loss=True,
                     want_grad=True):
        """returns {name: 0-d/1-element device tensor} with the reference's out_dict keys (lm_loss / matched_loss / obj_loss,
        feat_loss / qa_loss)."""
        assert task in ("vis_mask", "word_mask", "matched", "qa"), task
        lh = self.lang_heads
        self._task_run = task
        out = {}
        if task == "vis_mask":
            use_rows = want_grad and self.compact_head and self.has_vmask and 0 < self.n_mrows < self.MV
            self._hrows_step = (self.mrows, self.n_mrows) if use_rows else None
            self.encoder_forward(want_pooled=self.task_qa, ffn_rows=self._hrows_step)
            self._hrows = self._hrows_step
            try:
                self.head_forward()
                losses = self.losses_forward_backward(want_grad, feat_loss)
            finally:
                self._hrows = None
            out["obj_loss"] = losses[0:1]
            if feat_loss:
                out["feat_loss"] = losses[1:2]
        elif task == "word_mask":
            wl = word_labels.clone()
            wl[wl < 0] = -100           # the reference's data code writes -1, its loss ignores -100: any negative = not masked
            lh.word_labels.copy_(wl, non_blocking=True)
            lh.set_rows(word_rows if want_grad else None)      # labelled positions (from the data loader): masked-row head
            self.encoder_forward(want_pooled=self.task_qa)
            lh.mlm_fwd(self.lang_final)
            out["lm_loss"] = lh.mlm_loss()[0:1]
        elif task == "matched":
            lh.matched_labels.copy_(matched_labels, non_blocking=True)
            self.encoder_forward(want_pooled=True)
            lh.rel_fwd(self.pooled)
            out["matched_loss"] = lh.rel_loss()[1:2]
        else:
            assert self.task_qa, "task 'qa' needs a model built with the QA head (num_answers > 0)"
            self.encoder_forward(want_pooled=True)
        if self._qa_forward(qa_labels):
            out["qa_loss"] = self.answer.loss[0:1]
        self._qa_run = "qa_loss" in out
        return out

    def task_backward(self):
        """backward of the branch task_forward ran last; ACCUMULATES into store.grad."""
        task, qa, lh, ans = self._task_run, self._qa_run, self.lang_heads, self.answer
        self.begin_backward()
        GA = self.GA
        if task == "vis_mask":
            self._hrows = self._hrows_step
            try:
                self.head_backward(self.vr(GA), report=not qa)
            finally:
                self._hrows = None
            if self.need_lang:          # the language side of the last cross layer exists: zero gradient ...
                self.ops.zero(self.glang(GA))
            if qa:                      # ... unless the QA branch reads pooled_output
                cls_rows, d_cls = self._cls_views(GA)
                ans.bwd(self.pooled, cls_rows, d_cls)
                self._ready_heads()
            self.encoder_backward(self.need_lang)
            return
        self.zero_out_grads(GA)
        cls_rows, d_cls = self._cls_views(GA)
        if task == "word_mask":
            lh.mlm_bwd(self.glang(GA))
            if qa:
                ans.bwd(self.pooled, cls_rows, d_cls, accumulate=True)     # the MLM gradient of the [CLS] rows is there
        elif task == "matched":
            extra = None
            if qa:
                ans.bwd_to_pooled(self.pooled)
                extra = ans.dpooled
            lh.rel_bwd(self.pooled, cls_rows, d_cls, extra_dpooled=extra)
        else:
            ans.bwd(self.pooled, cls_rows, d_cls)
        self._ready_heads()
        self.encoder_backward(True)

    def _task_step(self, task, **kw):
        out = self.task_forward(task, **kw)
        self.clear_grads()
        self.task_backward()
        return out

    def qa_forward_backward(self, qa_labels):
        """XLxmertForPretraining.forward(task='qa') + backward on a task_qa model (ref lxrt/modeling.py:154-210, 292-306):
        un-masked codebook features in, total_loss = qa_loss.  Returns the device loss buffer [1]."""
        return self._task_step("qa", qa_labels=qa_labels)["qa_loss"]

    def word_mask_forward_backward(self, word_labels, word_rows=None, qa_labels=None):
        """XLxmertForPretraining.forward(task='word_mask') + backward (ref lxrt/modeling.py:211-219): un-masked codebook
        features in (set_inputs(cluster_ids=..., vis_mask=None)), MLM loss over `word_labels` (negative = ignored)."""
        return self._task_step("word_mask", word_labels=word_labels, word_rows=word_rows, qa_labels=qa_labels)["lm_loss"]

    def matched_forward_backward(self, matched_labels, qa_labels=None):
        """XLxmertForPretraining.forward(task='matched') + backward (ref lxrt/modeling.py:221-229)."""
        return self._task_step("matched", matched_labels=matched_labels, qa_labels=qa_labels)["matched_loss"]

    def vqa_forward(self):
        """VQAModel.forward (ref tasks/vqa_model.py:22-72): real grid features -> encoder -> pooled_output -> answer head."""
        self.encoder_forward(want_pooled=True)
        return self.answer.fwd(self.pooled)

    def vqa_forward_backward(self, targets):
        """one VQA/GQA fine-tune forward + backward (ref tasks/vqa.py:166-189): BCEWithLogitsLoss(logit, target).backward().
        Gradients land in store.grad; returns the device loss buffer [1]."""
        ans = self.answer
        ans.targets.copy_(targets, non_blocking=True)
        self.vqa_forward()
        self.zero_accumulated_grads()
        loss = ans.loss_fwd_bwd(True)
        GA = self.GA
        self.zero_out_grads(GA)                      # only the [CLS] rows of the language output carry gradient
        cls_rows, d_cls = self._cls_views(GA)
        ans.bwd(self.pooled, cls_rows, d_cls)
        self._ready_heads()
        self.encoder_backward(True)
        return loss

    def nlvr2_forward_backward(self, labels):
        """one NLVR2 fine-tune forward + backward (ref tasks/nlvr2.py:72 CrossEntropyLoss over the 2-way logit of each
        statement; tasks/nlvr2_model.py:50-86: encoder over the flattened [2P] (statement, image) rows, pooled outputs of a
        statement's two images concatenated).  `labels` [P] int64.  Returns the device loss buffer [1]."""
        ans = self.answer
        assert ans is not None and ans.pair, "build the store with task='nlvr2'"
        ans.labels.copy_(labels.reshape(-1), non_blocking=True)
        self.vqa_forward()
        self.zero_accumulated_grads()
        loss = ans.ce_loss_fwd_bwd(True)
        GA = self.GA
        self.zero_out_grads(GA)
        cls_rows, d_cls = self._cls_views(GA)
        ans.bwd(self.pooled, cls_rows, d_cls)
        self._ready_heads()
        self.encoder_backward(True)
        return loss

    def predict_codes(self):
        """head -> softmax -> max over the codebook (ref tasks/imggen_model.py:229-235): (max prob, argmax) per row."""
        self.ops.ce_fwd_bwd(self.logits, None, None, None, None, self.row_lse, self.row_argmax, self.row_maxprob,
                            self.MV, self.K, self.K, self.Kp, 1.0)
        return self.row_maxprob, self.row_argmax

    # Sampling never needs the logits themselves, only softmax(-1).max(-1): on the bf16 path the codebook contraction ends in
    # the XL_EPI_ROWMAX epilogue (per row and 64-column segment: max, sum exp, argmax) and xl_rowmax_combine finishes the rows --
    # 42 MB of segment records instead of writing and re-reading the 655 MB fp32 [B*V, 10000] matrix.  The codebook is padded
    # to a multiple of 256 rows for it (zero centroids with bias -1e30: never the maximum, exp() = 0).
    def fused_predict_available(self):
        return (self.cdtype == torch.bfloat16 and self.MV % 256 == 0 and self.F % 8 == 0 and hasattr(self.ops, "rowmax_combine")
                and os.environ.get("XL_FUSED_PREDICT", "1") != "0")

    def _prepare_fused_predict(self):
        Kq = (self.K + 255) // 256 * 256
        if getattr(self, "_cent_pad", None) is None or self._cent_pad.shape[0] != Kq:
            self._cent_pad = torch.zeros(Kq, self.F, dtype=self.cdtype, device=self.dev)
            self._bias_pad = torch.full((Kq,), -1e30, dtype=torch.float32, device=self.dev)
            self._rowmax_ws = torch.zeros((Kq // 64) * self.MV * 4, dtype=torch.float32, device=self.dev)
        self._cent_pad[:self.K].copy_(self.store.centroids_c)          # (frozen, but set_centroids may have replaced it)
        self._bias_pad[:self.K].copy_(self.hd["bc"][0])

    def predict_codes_fused(self):
        """head_forward(want_logits=False) must have run: self.feat -> (max prob, argmax) per row, no logits in memory."""
        Kq = self._cent_pad.shape[0]
        self.ops.gemm(self.feat, self._cent_pad, None, self._bias_pad, None, self._rowmax_ws, self.MV, Kq, self.F, self.F, self.F, Kq,
                      epilogue=EPI_ROWMAX)
        self.ops.rowmax_combine(self._rowmax_ws, Kq // 64, self.MV, self.row_maxprob, self.row_argmax, self.row_lse)
        return self.row_maxprob, self.row_argmax

    def _predict_step(self, fused):
        if fused:
            self.head_forward(want_logits=False)
            self.predict_codes_fused()
        else:
            self.head_forward()
            self.predict_codes()

    def sample_codes_nar(self, n_steps=4, on_step=None):
        """Iterative Mask-Predict sampling (ref tasks/imggen_model.py:169-243) without a host round trip between steps:
        re-mask the lowest-confidence positions -> encoder -> codebook head -> softmax-max / argmax -> keep the predictions
        of the masked positions.  Text inputs come from set_inputs (cluster_ids / vis_mask there are placeholders).
        Returns (code_ids [B,V] int64, code features [B*V, F] in the compute dtype, pred_prob [B*V] fp32); the caller
        hands `code.view(B,V,F).permute(0,2,1).view(B,F,g,g)` to the frozen GAN generator (stock PyTorch, ref :254).
        on_step(i): called after step i's update (return_intermediate of the reference, :245-248: materialise_codes() gives the
        code tensor of that moment)."""
        ops, B, V = self.ops, self.B, self.V
        st = self.store
        self.use_codebook, self.has_vmask = True, True
        self.cid.zero_()
        fused = self.fused_predict_available()
        if fused:
            self._prepare_fused_predict()
        for i in range(n_steps):
            n_mask = int((n_steps - i) / n_steps * V)                      # ref :201-202 (host arithmetic on the step index)
            if i == 0:
                self.vmask.fill_(1)                                        # ref :204-206
            else:
                ops.remask_lowest(self.row_maxprob, self.vmask, B, V, n_mask)
            self._reuse_lang_stack = i > 0                                 # the text is the same in every refinement step
            try:
                self.encoder_forward(want_pooled=False)                    # codebook_gather == where(mask, mask_feat, vis_emb(ids))
            finally:
                self._reuse_lang_stack = False
            self._predict_step(fused)
            ops.sampler_update(self.row_argmax, self.vmask, self.cid, B * V)
            if on_step is not None:
                on_step(i)
        ops.codebook_gather(self.cid, None, st.centroids_c, st.view("mask_feat"), self.feats, self.MV, self.F)
        return self.cid, self.feats, self.row_maxprob

    def materialise_codes(self, masked=False):
        """the sampler's code tensor [B*V, F] (compute dtype) of this moment: centroid rows of the current code ids; masked=True
        (autoregressive loop): `mask_feat` rows where vis_mask is still set, as the reference's `code` holds them (ref :99-102)."""
        st = self.store
        self.ops.codebook_gather(self.cid, self.vmask if masked else None, st.centroids_c, st.view("mask_feat"), self.feats,
                                 self.MV, self.F)
        return self.feats

    def sample_codes_ar(self, n_steps=None, mode="confidence", positions=None, trace=None, on_step=None):
        """Autoregressive sampling (ref tasks/imggen_model.py:49-153): one grid position per image is filled per step --
        the most confident not-yet-visited one ("confidence", the reference's default), position i ("tlbr"), or the host's
        shuffled order popped from the end ("random", positions = that list).  Same device-resident state as
        sample_codes_nar; `trace` (a list) receives a copy of vis_mask after every step."""
        ops, B, V = self.ops, self.B, self.V
        st = self.store
        n_steps = V if n_steps is None else n_steps
        self.use_codebook, self.has_vmask = True, True
        self.cid.zero_()
        self.vmask.fill_(1)
        if not hasattr(self, "visited"):
            self.visited = torch.zeros(B, V, dtype=torch.uint8, device=self.dev)
        self.visited.zero_()
        positions = list(positions) if positions is not None else None
        fused = self.fused_predict_available()
        if fused:
            self._prepare_fused_predict()
        for i in range(n_steps):
            cur = -1
            if mode == "random":
                cur = positions.pop() % V
                self.vmask[:, cur] = 1                                     # ref :104-108 (re-visits beyond V steps)
            elif mode == "tlbr":
                cur = i
            self._reuse_lang_stack = i > 0
            try:
                self.encoder_forward(want_pooled=False)
            finally:
                self._reuse_lang_stack = False
            self._predict_step(fused)
            ops.sampler_ar_update(self.row_maxprob, self.row_argmax, self.visited, self.vmask, self.cid, B, V, cur)
            if trace is not None:
                trace.append(self.vmask.clone())
            if on_step is not None:
                on_step(i)
        ops.codebook_gather(self.cid, self.vmask, st.centroids_c, st.view("mask_feat"), self.feats, self.MV, self.F)
        return self.cid, self.feats, self.row_maxprob

    # ------------------------------------------------------------ backward
    def zero_accumulated_grads(self):
        """start of a training step's backward: clear the gradient buffer, defer the column reductions' second stages."""
        self.clear_grads()
        self.begin_backward()

    accumulate = False      # set by the trainer for the 2nd.. micro-batch of a gradient-accumulation window (--update_freq)

    def clear_grads(self):
        """gradient buffer := 0 before a step's backward -- skipped when the optimizer pass of the previous step has already
        cleared it (trainer drop_grads: xl_adamw zero_grad) and nothing has accumulated since."""
        st = self.store
        if not self.grad_is_zero and not self.accumulate:
            self.ops.zero(st.grad[st.n_mat:st.n_used])
        self.grad_is_zero = False

    def begin_backward(self):
        """start of a backward pass that ACCUMULATES into the gradient buffer (the nn.Module path: zeroing is the caller's
        zero_grad(), as with autograd): second stages of the column reductions deferred until encoder_backward ends."""
        self.defer_reductions(True)
        assert not self._held["v"] and not self._held["l"]
        self._gen = {"v": 0, "l": 0}
        self._ws_i = {"v": 0, "l": 0}
        self._gen_guard = {"v": {}, "l": {}}    # (the previous backward ended with wgrad_sync_all on both streams)
        self.grad_is_zero = False
        self._lane_lo = {"v": 0, "l": self.store.language_range()[0]}

    def head_backward(self, d_vis, report=True):
        """consumes dlogits/dfeat from losses_forward_backward; writes d(vision_output) into d_vis.  report=False: another
        head's backward (QA branch) still follows before the head gradients are final."""
        ops, d, MV, F, K = self.ops, self.d, self.MV, self.F, self.K
        ops.block = "head"
        M = self._head_rows()
        hd = self.hd
        ops.colsum(self.dlogits, hd["bc"][1], M, self.Kp, self.Kp, ws=self.ws_wide("codebook", self.Kp))      # pad columns are zero; the bias unit is padded
        dfeat = self.tmp("dfeat", MV, F)
        if self.with_feat_loss:
            ops.gemm(self.dlogits, self.store.centroids_c, dfeat, None, self.dfeat, None, M, F, K, self.Kp, F, F, ldr=F,
                     a_kmajor=1, b_kmajor=0, epilogue=EPI_RESIDUAL)
        else:
            ops.gemm(self.dlogits, self.store.centroids_c, dfeat, None, None, None, M, F, K, self.Kp, F, F, a_kmajor=1, b_kmajor=0)
        ops.colsum(dfeat, hd["bf"][1], M, F, F, ws=self.ws)
        # weight gradients contract over the rows: round a compacted row count up to the K tile with zero gradient rows
        Mk = min(MV, (M + 63) // 64 * 64)
        if Mk > M:
            ops.zero(dfeat[M:Mk])
        self.wgrad_defer(dfeat, self.t_y, hd["wf"][1], F, d, Mk, F, d, d)
        dty = self.tmp("dz", MV, d)
        ops.gemm(dfeat, hd["wf"][0], dty, None, None, None, M, d, F, F, d, d, a_kmajor=1, b_kmajor=0)
        dth = self.tmp("dctx", MV, d)
        ops.layernorm_bwd(dty, self.t_h, hd["gt"][0], self.t_mean, self.t_rstd, dth, hd["gt"][1], hd["bbt"][1], None, M, d, ws=self.ws)
        dtp = self.tmp("dzm", MV, d)
        ops.gelu_bwd(dth, self.t_pre, dtp, M * d)
        ops.colsum(dtp, hd["bt"][1], M, d, d, ws=self.ws)
        if Mk > M:
            ops.zero(dtp[M:Mk])
        hv = self._hvis if self._hrows is None else self._hvis_buf
        self.wgrad_defer(dtp, hv, hd["wt"][1], d, d, Mk, d, d, d)
        self.wgrad_flush()
        if self._hrows is None:
            ops.gemm(dtp, hd["wt"][0], d_vis, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
        else:                       # gradient of the masked rows, scattered into an otherwise zero d(vision_output)
            dvc = self.tmp("dvis_c", MV, d)
            ops.gemm(dtp, hd["wt"][0], dvc, None, None, None, M, d, d, d, d, d, a_kmajor=1, b_kmajor=0)
            if self._ffn_rows_run is not None:          # ... or handed over compact: the last visual feed-forward block's
                self._dvis_c = dvc[:M]                  # backward runs on these rows too (_encoder_backward); d_vis is not written
            else:
                ops.zero(d_vis)
                ops.scatter_rows(dvc, self._hrows[0], d_vis, M, d, d, d)
        if report:
            self._ready_heads()

    def encoder_backward(self, have_lang_grad=False):
        """d(outputs) are expected in GA ([lang ; vis] rows; the language rows are ignored unless have_lang_grad)."""
        with Engine._Seeded(self):
            self._encoder_backward(have_lang_grad)

    def _encoder_backward(self, have_lang_grad=False):
        cfg, st, ops, d = self.cfg, self.store, self.ops, self.d
        ML, MV = self.ML, self.MV
        GA, GB = self.GA, self.GB
        L_, V_ = self.lr, self.vr
        if self.packed and have_lang_grad:      # d(language_output), dense layout -> the packed rows (zero rows in the pad tail)
            ops.gather_rows(self.glang_pad, self.lrows, L_(GA), ML, d, d, d)
        for i in reversed(range(cfg.x_layers)):
            blk = self.x_layers[i]
            lang_on = blk["lang_on"] and (have_lang_grad or i < cfg.x_layers - 1)
            if blk["lang_on"] and not lang_on:
                raise RuntimeError("engine built with need_lang=True needs d(language_output)")
            if self.pair_blocks and blk["lang_on"] and blk["vis_on"]:
                # both sides on this stream, their dX contractions two per launch; the language side keeps its own lane (scratch
                # sets, column-sum regions, weight-gradient grouping on its companion stream)
                self.run_pair(blk["ffn_v"].bwd_steps(V_(GA), V_(GB)), blk["ffn_l"].bwd_steps(L_(GA), L_(GB)))
                self.run_pair(blk["sa_v"].bwd_steps(V_(GB), V_(GA)), blk["sa_l"].bwd_steps(L_(GB), L_(GA)))
                if i == 0 or self.grad_ready is not None:
                    self.fork()
                    with self._lang_side():
                        if i == 0:              # the language side of the cross layers is reported by the main range (below):
                            self.wgrad_flush(pair=True, force=True)      # nothing of it may stay held into the language stack
                        if self.grad_ready is not None:
                            self._flush_if_reporting()
                            self.wgrad_sync_all()
                    if self.grad_ready is not None:
                        self.join()
            else:
                if blk["lang_on"]:
                    self.fork()
                    with self.lang_stream():
                        blk["ffn_l"].bwd(L_(GA), L_(GB))
                        blk["sa_l"].bwd(L_(GB), L_(GA))
                        if i == 0:                  # the language side of the cross layers is reported by the main stream (below):
                            self.wgrad_flush(pair=True, force=True)      # nothing of it may stay held into the language stack
                        if self.grad_ready is not None:     # (the main stream reports this layer, language side included)
                            self._flush_if_reporting()
                            self.wgrad_sync_all()
                if blk["vis_on"]:
                    if blk["ffn_v"].rows is not None:       # the forward ran this block on a row subset (encoder_forward ffn_rows)
                        rows, n = self._ffn_rows_run
                        dvc, self._dvis_c = self._dvis_c, None
                        assert i == cfg.x_layers - 1 and blk["ffn_v"].rows == n
                        if dvc is None:
                            raise RuntimeError("this forward computed the vision output at the masked rows only (encoder_forward "
                                               "ffn_rows): its backward starts from the head's compact gradient (head_backward)")
                        ops.block = blk["ffn_v"].tag
                        ops.zero(V_(GB))                    # d(the block's input): zero on the rows it did not read
                        if dvc is not Engine._NO_VIS_GRAD:
                            dsc = self.tmp("dffn_c", MV, d)[:n]
                            blk["ffn_v"].bwd(dvc, dsc)
                            ops.block = blk["ffn_v"].tag
                            ops.scatter_rows(dsc, rows, V_(GB), n, d, d, d)
                    else:
                        blk["ffn_v"].bwd(V_(GA), V_(GB))
                    blk["sa_v"].bwd(V_(GB), V_(GA))
                if blk["lang_on"]:
                    self.join()
            blk["cross"].bwd(GA, GB)
            self._ready(f"bert.encoder.x_layers.{i}.")
            GA, GB = GB, GA
        # ---- the two single-modality stacks: the paired layers first (a visual and a language layer per step on this stream) ...
        pairs = self._stack_pairs()
        for iv, il in reversed(pairs):
            (sa_v, ffn_v), (sa_l, ffn_l) = self.vis_layers[iv], self.lang_layers[il]
            self.run_pair(ffn_v.bwd_steps(V_(GA), V_(GB)), ffn_l.bwd_steps(L_(GA), L_(GB)))
            self.run_pair(sa_v.bwd_steps(V_(GB), V_(GA)), sa_l.bwd_steps(L_(GB), L_(GA)))
            if iv == 0 and self.grad_ready is not None:
                self.wgrad_flush(pair=True, force=True)         # (see the unpaired visual stack below)
            self._ready(f"bert.encoder.r_layers.{iv}.")
            if self.grad_ready is not None:
                self.fork()                 # (the language lane's fused column sums were combined on this stream: _ready)
                with self._lang_side():
                    self._flush_if_reporting()
                    self._ready_lang(st.range_of(f"bert.encoder.layer.{il}.")[1])
        n_lang_alone, n_vis_alone = cfg.l_layers - len(pairs), cfg.r_layers - len(pairs)
        # ---- ... then what is left of the longer stack beside the feature encoder's / the embeddings' backward
        self.fork()
        with self.lang_stream():            # ---- language stack + embeddings (HF:191-214) on the side stream
            for i in reversed(range(n_lang_alone)):
                sa, ffn = self.lang_layers[i]
                ffn.bwd(L_(GA), L_(GB))
                sa.bwd(L_(GB), L_(GA))
                self._flush_if_reporting()
                self._ready_lang(st.range_of(f"bert.encoder.layer.{i}.")[1])
            self.wgrad_flush(pair=True, force=True)      # an odd layer left over
            self.wgrad_sync()                             # the scratch / workspace generation written below: its last reader is done
            e = "bert.embeddings"
            dy, MLd = L_(GA), self.MLd
            if self.packed:                       # back to the dense [B*L] rows of the embedding kernels (zero at [PAD] positions)
                dy = self.tmp("emb_dy", MLd, d)
                ops.zero(dy)
                ops.scatter_rows(L_(GA), self.lrows, dy, ML, d, d, d)
            if self.p_hid > 0:
                ops.dropout(dy, dy, MLd, d, d, d, self.p_hid, self.seed(0))
            dpre = self.tmp("emb_dz", MLd, d)     # not "dz": layer 0's weight-gradient group may still be reading it
            ops.layernorm_bwd(dy, self.emb_pre, st.view(e + ".LayerNorm.weight"), self.emb_mean, self.emb_rstd, dpre,
                              st.gview(e + ".LayerNorm.weight"), st.gview(e + ".LayerNorm.bias"), None, MLd, d, ws=self.ws)
            emb = getattr(self, "embeds_mode", False)
            if emb:                                   # d(inputs_embeds) instead of d(word_embeddings): see set_inputs
                ops.zero(self._emb_grad)
            ops.embed_bwd(dpre, self._emb_ids if emb else self.ids, self.tt,
                          self._emb_grad if emb else st.gview(e + ".word_embeddings.weight"),
                          st.gview(e + ".position_embeddings.weight"), st.gview(e + ".token_type_embeddings.weight"),
                          self.B, self.L, d, order=self._emb_order if emb else self.word_order, n_types=cfg.type_vocab_size)
            self.flush_reductions()
            self._ready_lang(st.language_range()[1], flush=True)
            self.wgrad_sync_all()
        # ---- relational (visual) stack
        for i in reversed(range(n_vis_alone)):
            sa, ffn = self.vis_layers[i]
            ffn.bwd(V_(GA), V_(GB))
            sa.bwd(V_(GB), V_(GA))
            # the last (odd) layer's weight gradients: at once when its range is reported to a gradient exchange; otherwise behind
            # the feature encoder's backward below -- launched first they take 216 CUs for 0.2-0.4 ms and the encoder's last three
            # kernels on the chain crawl beside them (visn_ln_bwd: 55 us alone, 333 us measured in the step, profiles/r04a)
            if i == 0 and self.grad_ready is not None:
                self.wgrad_flush(pair=True, force=True)
            self._ready(f"bert.encoder.r_layers.{i}.")
        # ---- visual feature encoder (HF:468-476) + codebook input
        v = "bert.encoder.visn_fc"
        ops.block = "visn_fc"
        self.wgrad_sync()                     # (no block's bwd() follows the last _advance_gen: wait for this generation's guard here)
        dxv = self.tmp("dctx", MV, d)
        if self.p_hid > 0:
            ops.dropout(V_(GA), V_(GA), MV, d, d, d, self.p_hid, self.seed(1))
        ops.visn_ln_bwd(V_(GA), self.xv, self.pos, st.view(v + ".box_fc.weight"), st.view(v + ".box_fc.bias"),
                        st.view(v + ".visn_layer_norm.weight"), st.view(v + ".box_layer_norm.weight"), *self.vn_stats,
                        dxv, st.gview(v + ".visn_layer_norm.weight"), st.gview(v + ".visn_layer_norm.bias"),
                        st.gview(v + ".box_layer_norm.weight"), st.gview(v + ".box_layer_norm.bias"),
                        st.gview(v + ".box_fc.weight"), st.gview(v + ".box_fc.bias"), st.gview(v + ".visn_fc.bias"),
                        MV, d, self.P, ws=self.ws)
        if self.grad_ready is None:
            self.wgrad_flush(pair=True, force=True)          # (the visual stack's last held layer: see above)
        ops.gemm(dxv, self.feats, st.gview(v + ".visn_fc.weight"), None, None, None, d, self.F, MV, d, self.F, self.F,
                 a_kmajor=0, b_kmajor=0, out_f32=True, accumulate=1)      # (a K-split launch: into the cleared buffer, like every split one)
        if self.use_codebook and self.has_vmask:
            # d(mask_feat) = (sum over masked rows of d(xv)) W_visn   (ref lxrt/modeling.py:190-193: mask_feat is a Parameter)
            ops.zero(self.mf_tmp)
            ops.masked_colsum(dxv, self.vmask, self.mf_tmp, MV, d, d, ws=self.ws)
            self.flush_reductions()              # consumed right away
            ops.cast_from_f32(self.mf_tmp, self.mf_tmp_c, d)
            ops.gemm(self.mf_tmp_c, st.cview(v + ".visn_fc.weight"), st.gview("mask_feat"), None, None, None, 1, self.F, d,
                     d, self.F, self.F, a_kmajor=1, b_kmajor=0, out_f32=True, accumulate=1)
        self.flush_reductions()
        self.wgrad_sync_all()
        self.join()                          # language-stack gradients are final from here on
        self.defer_reductions(False)
        if self.grad_ready is not None:
            lo, hi = st.language_range()
            self._report("v", lo, flush=True)        # (already there: the last visual layer reported it)
            self._lane_lo["v"] = hi
            self._report("v", st.n_used, flush=True)

    # ------------------------------------------------------------ whole vis_mask step (forward + backward)
    def vis_mask_forward_backward(self, feat_loss=True, qa_labels=None):
        """XLxmertForPretraining.forward(task='vis_mask') + loss.backward() (ref lxrt/modeling.py:154-308,
        lxmert_pretrain.py:338).  Gradients land in store.grad; returns the device loss buffer [obj_loss, feat_loss]
        (a task_qa model's qa_loss is in self.answer.loss)."""
        self._task_step("vis_mask", feat_loss=feat_loss, qa_labels=qa_labels)
        return self.losses
