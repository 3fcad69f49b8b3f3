"""Tensor-level wrappers over the C ABI (include/xlxmert_hip.h).

`HipOps` hands raw device pointers of torch tensors to libxlxmert_hip.so on torch's current HIP stream.
torch is only the allocator / stream provider here.  Every method mirrors one `xl_*` entry point
(same argument meaning); tensors may be views -- the pointer passed is `tensor.data_ptr()`.
There is no CPU or eager fallback: a CPU tensor or a missing library raises.
"""
import threading

import torch

from ._lib import XlError, get_lib

XL_F32, XL_BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_TANH, EPI_ROWMAX, EPI_GELU_DG, EPI_MULAUX = 0, 1, 2, 3, 4, 5, 6, 7

TORCH_DTYPE = {XL_F32: torch.float32, XL_BF16: torch.bfloat16}
_SLAB_WS = {}


def xl_dtype(torch_dtype):
    if torch_dtype == torch.float32:
        return XL_F32
    if torch_dtype == torch.bfloat16:
        return XL_BF16
    raise XlError(f"unsupported dtype {torch_dtype}: the HIP path computes in float32 or bfloat16")


class GemmCall:
    """arguments of one HipOps.gemm call, held back so that a caller can hand two of them to gemm_pair (engine: paired blocks)"""
    __slots__ = ("a", "kw", "tag")

    def __init__(self, *a, tag=None, **kw):
        self.a, self.kw, self.tag = a, kw, tag          # tag: label of the issuing model block (HipOps.block, bench.py's per-block table)


class HipOps:
    """One instance per compute dtype."""

    def __init__(self, dtype):
        self.lib = get_lib()
        self.dtype = dtype
        self.dt = xl_dtype(dtype)
        self.ncalls = 0              # C-ABI calls issued through this object (engine.run_pair: did a block's step launch anything?)
        self.block = ""              # label of the model block issuing the current calls (set by the engine; bench.py's per-block table)
        # this object's library context (include/xlxmert_hip.h xl_ctx_*): step-seed pointer, deferred reductions, slab
        # workspaces and kernel switches set through it are invisible to every other HipOps of the process
        self._slab_registered = set()
        self.ctx = int(self.lib.raw("xl_ctx_create")())
        if self.ctx <= 0:
            raise XlError("xl_ctx_create failed")

    # The context bound to the CALLING THREAD, mirrored on the host so that a call binds only when it has to.  xl_ctx_bind is
    # thread-local in the library (csrc/common.h t_ctx), so the mirror is too: autograd runs the backward of the nn.Module
    # surface (_VqaFn / _HeadFn / _EncoderFn) on a device worker thread, which starts in the default context whatever the main
    # thread has bound.
    _tls = threading.local()

    @staticmethod
    def bound():
        return getattr(HipOps._tls, "ctx", None)

    @staticmethod
    def forget_binding():
        """the calling thread's binding is unknown (a launch plan was replayed: its last recorded bind is in effect)"""
        HipOps._tls.ctx = None

    def _call(self, name, *args):
        if HipOps.bound() != self.ctx:
            self.lib.call("xl_ctx_bind", self.ctx)
            HipOps._tls.ctx = self.ctx
        self.ncalls += 1
        return self.lib.call(name, *args)

    def _call_exp(self, name, value, default):
        """setter of an EXPERIMENTAL switch (include/xlxmert_hip.h, section EXPERIMENTAL): with the default library, asking for the
        default value is a no-op (the variant does not exist there), anything else raises through Lib.call"""
        if not self.lib.experimental and int(value) == default:
            return 0
        return self._call(name, int(value))

    def rebind(self):
        """bind this object's context unconditionally (first entry of a recorded launch plan; after a plan replay, whose last
        recorded bind -- possibly another object's -- is the one in effect)."""
        HipOps._tls.ctx = self.ctx
        self.lib.call("xl_ctx_bind", self.ctx)

    def __del__(self):
        try:                             # raw calls: a collection that happens while a plan is being recorded must not end up in it
            if HipOps.bound() == self.ctx:
                self.lib.raw("xl_ctx_bind")(0)
                HipOps._tls.ctx = None
            self.lib.raw("xl_ctx_destroy")(self.ctx)
        except Exception:
            pass

    # -- plumbing
    @staticmethod
    def _p(t):
        if t is None:
            return None
        if not t.is_cuda:
            raise XlError("HipOps got a CPU tensor: the X-LXMERT hot path has no CPU fallback")
        return t.data_ptr()

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def set_lds_transpose_read(self, enable):
        self._call("xl_set_lds_transpose_read", int(enable))

    def set_gemm_pingpong(self, mode):
        """0: 128x128 GEMM kernel only; 1: by shape (default); 2: 256x256 ping-pong kernel whenever eligible."""
        self._call("xl_set_gemm_pingpong", int(mode))

    def set_gemm_persistent(self, on):
        """persistent ping-pong kernel for multi-round, short-K launches: 1 = when eligible, 0 = never (default; env
        XL_GEMM_PERSIST sets the initial value)."""
        self._call_exp("xl_set_gemm_persistent", on, 0)

    def set_gemm_duo(self, mode):
        """128x192 tiles, two four-wave workgroups per CU: 0 never, 1 small launches (default), 2 every eligible launch."""
        self._call("xl_set_gemm_duo", int(mode))

    def set_gemm_q(self, mode):
        """128x192 tiles by eight 128-register waves, two workgroups per CU: 0 never, 1 short contractions, 2 every eligible launch."""
        self._call_exp("xl_set_gemm_q", mode, 0)

    def set_gemm_relay(self, mode):
        """0 never / 1 multi-round K <= 1024 launches / 2 every eligible launch on the role-trading persistent kernel (gemm_relay.hip)"""
        self._call_exp("xl_set_gemm_relay", mode, 0)

    def set_gemm_relay_wgs(self, wgs):
        self._call_exp("xl_set_gemm_relay_wgs", wgs, 256)

    def set_gemm_split_epi(self, on):
        """K split of few-tile, deep-K launches with an epilogue through the stream's slab workspace: 0 never (default), 1 when eligible."""
        self._call_exp("xl_set_gemm_split_epi", on, 0)

    def set_gemm_tail_split(self, max_tail_tiles, min_k):
        self._call("xl_set_gemm_tail_split", int(max_tail_tiles), int(min_k))

    def gemm_trace(self, buffer):
        """device tensor for the ping-pong kernel's per-workgroup time stamps, or None (xl_gemm_trace)."""
        self._call("xl_gemm_trace", self._p(buffer))

    def set_gemm_tile192(self, mode):
        """0: 256x256 tiles only; 1: 256x192 where it shortens the launch (default); 2: whenever eligible."""
        self._call_exp("xl_set_gemm_tile192", mode, 0)

    # -- stream plumbing of a step as C-ABI calls (so that a recorded launch plan contains them: _lib.LaunchPlan)
    def zero(self, t):
        """t.zero_() on the current stream (t contiguous)."""
        assert t.is_contiguous()
        self._call("xl_memset", self._p(t), 0, t.numel() * t.element_size(), self._stream())

    _events, _ev_next = [], 0

    def stream_fork(self, from_stream, to_stream):
        """`to_stream` continues after everything queued on `from_stream` so far (event record + stream wait).  Events come
        from a process-wide ring: a wait refers to the record that precedes it at enqueue time, so re-recording is safe."""
        cls = HipOps
        if len(cls._events) < 512:
            ev = int(self.lib.raw("xl_event_create")())
            if not ev:
                raise XlError("xl_event_create failed")
            cls._events.append(ev)
        else:
            ev = cls._events[cls._ev_next % 512]
        cls._ev_next += 1
        self._call("xl_stream_fork", ev, from_stream.cuda_stream, to_stream.cuda_stream)

    def new_event(self):
        ev = int(self.lib.raw("xl_event_create")())
        if not ev:
            raise XlError("xl_event_create failed")
        return ev

    def event_record(self, ev, stream):
        self._call("xl_event_record", ev, stream.cuda_stream)

    def stream_wait(self, ev, stream):
        self._call("xl_stream_wait", ev, stream.cuda_stream)

    def set_step_seed_ptr(self, step_seed):
        """device tensor (one int64 >= 0) holding the step part of every dropout seed, or None (xl_set_step_seed_ptr)."""
        self._call("xl_set_step_seed_ptr", self._p(step_seed))

    def set_deferred_reduce(self, on):
        self._call("xl_set_deferred_reduce", int(on))

    def flush_reductions(self):
        self._call("xl_flush_reductions", self._stream())

    def flush_reductions_on(self, producer_stream):
        """combine everything pending from `producer_stream` with a launch on the CURRENT stream (ordered after the producers by the caller)"""
        self._call("xl_flush_reductions_on", producer_stream.cuda_stream, self._stream())

    # -- dense contractions
    def gemm(self, A, B, C, bias, residual, aux, M, N, K, lda, ldb, ldc, ldr=0, ldx=0, a_kmajor=1, b_kmajor=1,
             out_f32=False, epilogue=EPI_NONE, alpha=1.0, accumulate=0, p_drop=0.0, seed=0, colsum=None, ws=None):
        """colsum: optional fp32 [N] that receives += the column sums of C (bias gradient), with workspace `ws`."""
        self._call("xl_gemm", self._p(A), self._p(B), self._p(C), self._p(bias), self._p(residual), self._p(aux),
                      M, N, K, lda, ldb, ldc, ldr, ldx, int(a_kmajor), int(b_kmajor), self.dt,
                      XL_F32 if out_f32 else self.dt, epilogue, float(alpha), int(accumulate), float(p_drop),
                      int(seed), self._p(colsum), self._p(ws if colsum is not None else None), self._stream())

    @staticmethod
    def _gemm_args(A, B, C, bias, residual, aux, M, N, K, lda, ldb, ldc, ldr=0, ldx=0, a_kmajor=1, b_kmajor=1,
                   out_f32=False, epilogue=EPI_NONE, alpha=1.0, accumulate=0, p_drop=0.0, seed=0, colsum=None, ws=None):
        return dict(A=A, B=B, C=C, bias=bias, residual=residual, aux=aux, M=M, seed=int(seed), colsum=colsum, ws=ws,
                    shared=(N, K, lda, ldb, ldc, ldr, ldx, int(a_kmajor), int(b_kmajor), bool(out_f32), epilogue, float(alpha),
                            int(accumulate), float(p_drop)))

    def gemm_pair(self, c0, c1):
        """two gemm() calls, c = GemmCall(*args, **kwargs) each, of one shape class (same N, K, leading dimensions, layouts,
        epilogue, alpha, dropout probability; bf16 output, no accumulation): ONE launch when the library can pair them
        (xl_gemm_pair), the two launches otherwise.  Same results either way."""
        g0, g1 = self._gemm_args(*c0.a, **c0.kw), self._gemm_args(*c1.a, **c1.kw)
        sh = g0["shared"]
        if sh != g1["shared"] or sh[9] or sh[12]:
            self.gemm(*c0.a, **c0.kw)
            self.gemm(*c1.a, **c1.kw)
            return
        N, K, lda, ldb, ldc, ldr, ldx, ak, bk, _, epi, alpha, _, p_drop = sh
        per = []
        for g in (g0, g1):
            cs = g["colsum"]
            per += [self._p(g["A"]), self._p(g["B"]), self._p(g["C"]), self._p(g["bias"]), self._p(g["residual"]), self._p(g["aux"]),
                    g["M"], g["seed"], self._p(cs), self._p(g["ws"] if cs is not None else None)]
        self._call("xl_gemm_pair", *per, N, K, lda, ldb, ldc, ldr, ldx, ak, bk, self.dt, self.dt, epi, alpha, p_drop, self._stream())

    def set_gemm_pair(self, on):
        """two-problem launches (xl_gemm_pair): 1 when eligible (default), 0 always two launches."""
        self._call_exp("xl_set_gemm_pair", on, 1)

    def gemm_wgrad_group(self, problems, overwrite_mask=0):
        """problems: list of (dY [K, M], X [K, N], dW [M, N] fp32, M, N, K, lda, ldb, ldc): dW += dY^T X, one launch; bit i of
        overwrite_mask: dW_i = dY_i^T X_i instead (a weight with one gradient contribution per step)."""
        import ctypes
        n = len(problems)
        vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
        cols = list(zip(*problems))
        ptrs = [vp(*[self._p(t) for t in cols[j]]) for j in range(3)]
        ints = [ia(*[int(v) for v in cols[j]]) for j in range(3, 9)]
        self._call("xl_gemm_wgrad_group", *ptrs, *ints, n, int(overwrite_mask), self.dt, self._stream())

    def wgrad_group_one_writer(self, problems):
        """True when the grouped launch of `problems` gives every output tile one writer (no K split) and every tile is whole:
        the launch then honours an overwrite_mask with plain stores (otherwise by clearing C first)."""
        import ctypes
        n = len(problems)
        ia = ctypes.c_int * n
        M, N, K = (ia(*[int(pr[j]) for pr in problems]) for j in (3, 4, 5))
        if self.dt != XL_BF16:
            return True                      # (fp32: one plain xl_gemm per problem, never split)
        one = int(self.lib.raw("xl_gemm_wgrad_group_splitk")(ctypes.cast(M, ctypes.c_void_p), ctypes.cast(N, ctypes.c_void_p),
                                                             ctypes.cast(K, ctypes.c_void_p), n)) == 1
        return one and all(pr[3] % 256 == 0 and pr[4] % 256 == 0 for pr in problems)

    def set_gemm_wgrad_slabs(self, on):
        """weight-gradient K splits through the slab workspace (fixed summation order) instead of fp32 atomics."""
        self._call("xl_set_gemm_wgrad_slabs", int(on))

    def gemm_workspace(self, slabs=256, stream=None):
        """allocate and register the slab workspace of `stream` (default: the current one): xl_gemm_set_workspace."""
        st = stream if stream is not None else torch.cuda.current_stream()
        key = (st.device_index, st.cuda_stream, int(slabs))
        nbytes = int(self.lib.raw("xl_gemm_workspace_bytes")(int(slabs)))
        if key not in _SLAB_WS:               # the memory: one per stream for the life of the process (launches on one stream run
            _SLAB_WS[key] = torch.zeros(nbytes // 4, dtype=torch.float32, device=torch.device("cuda", st.device_index))   # in order)
        if key not in self._slab_registered:  # the registration: per context
            self._call("xl_gemm_set_workspace", _SLAB_WS[key].data_ptr(), nbytes, st.cuda_stream)
            self._slab_registered.add(key)
        return _SLAB_WS[key]

    # -- LayerNorm family
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, M, N, eps):
        self._call("xl_layernorm_fwd", self._p(x), self._p(gamma), self._p(beta), self._p(y), self._p(mean),
                      self._p(rstd), M, N, float(eps), self.dt, self._stream())

    def workspace_floats(self, N):
        return int(self.lib.raw("xl_workspace_floats")(int(N)))

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias_prev, M, N, ws=None, dx_dropped=None,
                      p_drop=0.0, seed=0):
        self._call("xl_layernorm_bwd", self._p(dy), self._p(x), self._p(gamma), self._p(mean), self._p(rstd),
                      self._p(dx), self._p(dgamma), self._p(dbeta), self._p(dbias_prev), M, N, self._p(ws),
                      self._p(dx_dropped), float(p_drop), int(seed), self.dt, self._stream())

    def visn_ln_fwd(self, xv, pos, wbox, bbox, gv, bv, gb, bb, y, mean_v, rstd_v, mean_b, rstd_b, M, N, P, eps):
        self._call("xl_visn_ln_fwd", self._p(xv), self._p(pos), self._p(wbox), self._p(bbox), self._p(gv),
                      self._p(bv), self._p(gb), self._p(bb), self._p(y), self._p(mean_v), self._p(rstd_v),
                      self._p(mean_b), self._p(rstd_b), M, N, P, float(eps), self.dt, self._stream())

    def visn_ln_bwd(self, dy, xv, pos, wbox, bbox, gv, gb, mean_v, rstd_v, mean_b, rstd_b, dxv, dgv, dbv, dgb, dbb,
                    dwbox, dbbox, dbias_visn, M, N, P, ws=None):
        self._call("xl_visn_ln_bwd", self._p(dy), self._p(xv), self._p(pos), self._p(wbox), self._p(bbox),
                      self._p(gv), self._p(gb), self._p(mean_v), self._p(rstd_v), self._p(mean_b), self._p(rstd_b),
                      self._p(dxv), self._p(dgv), self._p(dbv), self._p(dgb), self._p(dbb), self._p(dwbox),
                      self._p(dbbox), self._p(dbias_visn), M, N, P, self._p(ws), self.dt, self._stream())

    # -- embeddings / codebook
    def embed_ln_fwd(self, ids, tt, word, pos, type_, gamma, beta, y, pre, mean, rstd, B, L, N, eps):
        self._call("xl_embed_ln_fwd", self._p(ids), self._p(tt), self._p(word), self._p(pos), self._p(type_),
                      self._p(gamma), self._p(beta), self._p(y), self._p(pre), self._p(mean), self._p(rstd), B, L, N,
                      float(eps), self.dt, self._stream())

    def embed_bwd(self, dpre, ids, tt, dword, dpos, dtype_tab, B, L, N, order=None, n_types=2):
        """order: int32 [B*L] rows sorted by (id, row) -- trainer.word_order_of(input_ids); None: the (slow) scanning kernel"""
        self._call("xl_embed_bwd", self._p(dpre), self._p(ids), self._p(tt), self._p(order), self._p(dword), self._p(dpos),
                      self._p(dtype_tab), B, L, N, int(n_types), self.dt, self._stream())

    def codebook_gather(self, cluster_ids, vis_mask, centroids, mask_feat, feats, M, F):
        self._call("xl_codebook_gather", self._p(cluster_ids), self._p(vis_mask), self._p(centroids),
                      self._p(mask_feat), self._p(feats), M, F, self.dt, self._stream())

    def masked_colsum(self, x, mask, out, M, N, ldx, ws=None):
        self._call("xl_masked_colsum", self._p(x), self._p(mask), self._p(out), M, N, ldx, self._p(ws), self.dt,
                      self._stream())

    def colsum(self, x, out, M, N, ldx, ws=None):
        self._call("xl_colsum", self._p(x), self._p(out), M, N, ldx, self._p(ws), self.dt, self._stream())

    def dropout(self, x, y, M, N, ldx, ldy, p_drop, seed):
        self._call("xl_dropout", self._p(x), self._p(y), M, N, ldx, ldy, float(p_drop), int(seed), self.dt,
                      self._stream())

    def gelu_bwd(self, dy, pre, dx, n):
        self._call("xl_gelu_bwd", self._p(dy), self._p(pre), self._p(dx), n, self.dt, self._stream())

    def tanh_bwd(self, dy, y, dx, n):
        self._call("xl_tanh_bwd", self._p(dy), self._p(y), self._p(dx), n, self.dt, self._stream())

    def bce_logits_fwd_bwd(self, logits, targets, dlogits, loss, M, N, ld_logits, ld_targets, ld_dlogits):
        self._call("xl_bce_logits_fwd_bwd", self._p(logits), self._p(targets), self._p(dlogits), self._p(loss), M, N,
                      ld_logits, ld_targets, ld_dlogits, self.dt, self._stream())

    def remask_lowest(self, prob, vis_mask, B, V, n_mask):
        self._call("xl_remask_lowest", self._p(prob), self._p(vis_mask), B, V, int(n_mask), self._stream())

    def sampler_update(self, pred_ids, vis_mask, code_ids, n):
        self._call("xl_sampler_update", self._p(pred_ids), self._p(vis_mask), self._p(code_ids), n, self._stream())

    def sampler_ar_update(self, prob, pred_ids, visited, vis_mask, code_ids, B, V, fixed_pos=-1):
        self._call("xl_sampler_ar_update", self._p(prob), self._p(pred_ids), self._p(visited), self._p(vis_mask),
                      self._p(code_ids), B, V, int(fixed_pos), self._stream())

    # -- attention core
    def sdpa_keep_bits_bytes(self, B, H, nq, nk, dh):
        """bytes of the buffer in which sdpa_fwd leaves its dropout decisions for sdpa_bwd (0: this geometry / dtype runs on kernels
        that evaluate the mask hash in both directions)"""
        return int(self.lib.raw("xl_sdpa_keep_bits_bytes")(B, H, nq, nk, dh, self.dt))

    def sdpa_fwd(self, q, k, v, key_mask, o, lse, B, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop=0.0, seed=0,
                 q_off=None, k_off=None, q_pad=0, k_pad=0, keep_bits=None):
        """q_off / k_off: int32 [B+1] row offsets of a PACKED side (None = dense [B, n] rows); q_pad: rows [q_off[B], q_pad) of
        `o` are written as zeros.  keep_bits: int32 buffer of sdpa_keep_bits_bytes(...) bytes that receives the dropout decisions."""
        self._call("xl_sdpa_fwd", self._p(q), self._p(k), self._p(v), self._p(key_mask), self._p(o), self._p(lse),
                   B, H, nq, nk, dh, ldq, ldk, ldv, ldo, float(scale), float(p_drop), int(seed), self._p(q_off), self._p(k_off),
                   int(q_pad), int(k_pad), self._p(keep_bits), self.dt, self._stream())

    def sdpa_bwd(self, q, k, v, key_mask, dout, lse, dq, dk, dv, B, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq, lddk,
                 lddv, scale, p_drop=0.0, seed=0, bias_grad=None, ws=None, q_off=None, k_off=None, q_pad=0, k_pad=0, keep_bits=None):
        self._call("xl_sdpa_bwd", self._p(q), self._p(k), self._p(v), self._p(key_mask), self._p(dout),
                   self._p(lse), self._p(dq), self._p(dk), self._p(dv), B, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq,
                   lddk, lddv, float(scale), float(p_drop), int(seed), self._p(bias_grad), self._p(ws), self._p(q_off),
                   self._p(k_off), int(q_pad), int(k_pad), self._p(keep_bits), self.dt, self._stream())

    def attn_probs(self, q, k, key_mask, lse, probs, B, H, nq, nk, dh, ldq, ldk, scale, p_drop=0.0, seed=0, q_off=None, k_off=None):
        """probs fp32 [B, H, nq, nk] := softmax (after dropout) of one attention block, from q, k and the saved lse."""
        self._call("xl_attn_probs", self._p(q), self._p(k), self._p(key_mask), self._p(lse), self._p(probs), B, H, nq, nk, dh,
                   ldq, ldk, float(scale), float(p_drop), int(seed), self._p(q_off), self._p(k_off), self.dt, self._stream())

    # -- head losses
    def mask_counts(self, labels, vis_mask, counts, nmask, B, V):
        self._call("xl_mask_counts", self._p(labels), self._p(vis_mask), self._p(counts), self._p(nmask), B, V,
                      self._stream())

    def ce_fwd_bwd(self, logits, labels, counts, dlogits, loss_out, row_lse, row_argmax, row_maxprob, M, K, ldl, lddl,
                   grad_scale=1.0):
        self._call("xl_ce_fwd_bwd", self._p(logits), self._p(labels), self._p(counts), self._p(dlogits),
                      self._p(loss_out), self._p(row_lse), self._p(row_argmax), self._p(row_maxprob), M, K, ldl, lddl,
                      float(grad_scale), self.dt, self._stream())

    def featloss_fwd_bwd(self, pred, centroids, cluster_ids, vis_mask, nmask, dpred, loss_out, B, V, F,
                         grad_scale=1.0, rows=None, n_rows=0, targets=None):
        """targets: optional [B*V, F] regression targets (label_dict['feat_labels']); default = centroids[cluster_ids]."""
        self._call("xl_featloss_fwd_bwd", self._p(pred), self._p(centroids), self._p(cluster_ids),
                      self._p(vis_mask), self._p(nmask), self._p(dpred), self._p(loss_out), B, V, F, float(grad_scale),
                      self._p(rows), int(n_rows), self._p(targets), self.dt, self._stream())

    def gather_rows(self, src, rows, dst, n_rows, N, ld_src, ld_dst):
        self._call("xl_gather_rows", self._p(src), self._p(rows), self._p(dst), n_rows, N, ld_src, ld_dst, self.dt, self._stream())

    def scatter_rows(self, src, rows, dst, n_rows, N, ld_src, ld_dst):
        self._call("xl_scatter_rows", self._p(src), self._p(rows), self._p(dst), n_rows, N, ld_src, ld_dst, self.dt, self._stream())

    def rowmax_combine(self, ws, n_seg, M, row_maxprob, row_argmax, row_lse=None):
        """second half of gemm(epilogue=EPI_ROWMAX, aux=ws): per-row argmax / max softmax probability / log-sum-exp."""
        self._call("xl_rowmax_combine", self._p(ws), n_seg, M, self._p(row_maxprob), self._p(row_argmax), self._p(row_lse),
                      self._stream())

    def gather_labels(self, labels, rows, out, n_rows):
        self._call("xl_gather_labels", self._p(labels), self._p(rows), self._p(out), n_rows, self._stream())

    def sumsq_scratch(self, device):
        """zeroed scratch for xl_sumsq (block partials + ticket): one per call site that may run concurrently with another."""
        n = int(self.lib.raw("xl_sumsq_scratch_bytes")())
        return torch.zeros((n + 3) // 4, dtype=torch.float32, device=device)

    def sumsq(self, g, out, n, scratch):
        self._call("xl_sumsq", self._p(g), self._p(out), n, self._p(scratch), self._stream())

    def schedule_step(self, step, base_lr, warmup_steps, total_steps, beta1, beta2, lr_and_steps):
        """device-side: step[0] += 1; lr_and_steps = {lr(t), 1-beta1^t, 1-beta2^t, t} (stream-ordered before adamw)."""
        self._call("xl_schedule_step", self._p(step), float(base_lr), int(warmup_steps), int(total_steps), float(beta1),
                      float(beta2), self._p(lr_and_steps), self._stream())

    def adamw(self, p, g, m, v, p_compute, decay_flags, sumsq, lr_and_steps, n, beta1, beta2, eps, weight_decay,
              max_norm, grad_scale=1.0, chunk_steps=None, zero_grad=False):
        self._call("xl_adamw", self._p(p), self._p(g), self._p(m), self._p(v), self._p(p_compute),
                      self._p(decay_flags), self._p(chunk_steps), self._p(sumsq), self._p(lr_and_steps), n, float(beta1), float(beta2),
                      float(eps), float(weight_decay), float(max_norm), float(grad_scale), int(zero_grad), self.dt, self._stream())

    # -- gradient exchange behind the C ABI (csrc/comm.hip): RCCL collectives as plan-able calls
    def comm_allreduce(self, comm, buf, n, after_stream=None):
        """buf[:n] := sum over ranks (in place) on the communicator's stream, after everything queued so far on `after_stream`
        (default: the current stream)."""
        st = after_stream.cuda_stream if after_stream is not None else self._stream()
        self._call("xl_comm_allreduce", int(comm), self._p(buf), int(n), xl_dtype(buf.dtype), st)

    def comm_reduce_scatter(self, comm, buf, n, rank, world, after_stream=None):
        """in place over buf[:n] (n a multiple of `world`): rank r's piece buf[r*n/world : (r+1)*n/world] := sum over ranks of
        that piece; the other pieces of buf are left with partial sums (RCCL's in-place reduce-scatter layout)."""
        per = int(n) // int(world)
        st = after_stream.cuda_stream if after_stream is not None else self._stream()
        self._call("xl_comm_reduce_scatter", int(comm), self._p(buf), self._p(buf) + rank * per * buf.element_size(), per,
                   xl_dtype(buf.dtype), st)

    def comm_allgather(self, comm, buf, n, rank, world, after_stream=None):
        """in place over buf[:n]: every rank ends up with every rank's piece buf[r*n/world : (r+1)*n/world]."""
        per = int(n) // int(world)
        st = after_stream.cuda_stream if after_stream is not None else self._stream()
        self._call("xl_comm_allgather", int(comm), self._p(buf) + rank * per * buf.element_size(), self._p(buf), per,
                   xl_dtype(buf.dtype), st)

    def comm_wait(self, comm, stream=None):
        st = stream.cuda_stream if stream is not None else self._stream()
        self._call("xl_comm_wait", int(comm), st)

    def take_f32(self, src, idx, own_lo, own_hi, dst):
        """dst[j] = src[idx[j]] where own_lo <= idx[j] < own_hi, else 0 (xl_take_f32: the sharded exchange's fp32 side car)"""
        self._call("xl_take_f32", self._p(src), self._p(idx), int(idx.numel()), int(own_lo), int(own_hi), self._p(dst), self._stream())

    def put_f32(self, dst, idx, src):
        self._call("xl_put_f32", self._p(dst), self._p(idx), int(idx.numel()), self._p(src), self._stream())

    def cast_from_f32(self, src, dst, n):
        self._call("xl_cast_from_f32", self._p(src), self._p(dst), n, self.dt, self._stream())

    def cast_to_f32(self, src, dst, n):
        self._call("xl_cast_to_f32", self._p(src), self._p(dst), n, self.dt, self._stream())
