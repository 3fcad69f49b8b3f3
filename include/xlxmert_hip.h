/*
 * xlxmert_hip.h -- C ABI of libxlxmert_hip.so: the MI355X (gfx950) kernels behind the
 * X-LXMERT hot path (LxmertEncoder stack + masked-visual-token head, data-parallel training step).
 *
 * The reference has no FFI seam on this path: the seam is the Python nn.Module API
 * (SURVEY.md section 8b).  Each entry point below replaces the aten op sequence the reference
 * launches at the cited site ("HF:" = transformers/models/lxmert/modeling_lxmert.py, the
 * un-vendored dependency where the arithmetic lives; "ref:" = /root/reference/...).
 *
 * Conventions
 *   - plain pointers + sizes only; no torch types.  All pointers are DEVICE pointers to
 *     caller-owned, 16-byte-aligned memory; the library never allocates tensor memory.
 *   - every call is asynchronous on the caller-supplied hipStream_t (passed as void*).
 *   - return 0 on success, negative XL_ERR_* otherwise; xl_last_error() gives the text.
 *   - dtype: activation / compute-weight element type.  XL_F32 = exact-fp32 path (parity
 *     config, fp32 FMA / fp32 accumulate); XL_BF16 = bf16 operands, fp32 accumulate on MFMA.
 *     LayerNorm statistics, softmax, losses, gradients of parameters, optimizer state: always fp32.
 *   - "ld*" are leading dimensions in ELEMENTS.
 *   - threading / state: the library keeps no process-global SETTINGS.  (Two process-wide tables exist and are not
 *     settings: the RCCL function table resolved by the first xl_comm_* call, immutable afterwards, and the table that maps xl_comm_*
 *     handles to communicators -- csrc/comm.hip g_rccl / g_comms, both mutex-guarded.)  Everything a caller can
 *     set -- the dropout step-seed pointer, the deferred-reduction switch and its pending lists, the per-stream slab
 *     workspaces, the kernel-choice / debug switches -- belongs to a CONTEXT (xl_ctx_create); a thread binds the context it
 *     works for with xl_ctx_bind and every xl_* call it makes afterwards reads that context.  Calls are re-entrant across
 *     streams and across contexts; a context bound by several threads serialises only its own two small maps.  A caller that
 *     never creates a context works in the default one (handle 0).  xl_last_error() is per thread.
 */
#ifndef XLXMERT_HIP_H
#define XLXMERT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XL_F32  0
#define XL_BF16 1

#define XL_OK             0
#define XL_ERR_BAD_SHAPE -1
#define XL_ERR_BAD_DTYPE -2
#define XL_ERR_UNALIGNED -3
#define XL_ERR_HIP       -4
#define XL_ERR_BAD_ARG   -5
#define XL_ERR_RCCL      -6

/* GEMM epilogues */
#define XL_EPI_NONE     0   /* C = acc (+bias)                                                      */
#define XL_EPI_GELU     1   /* aux = acc+bias (pre-activation, saved); C = gelu_erf(aux)   HF:325-328 */
#define XL_EPI_RESIDUAL 2   /* C = dropout(acc+bias) + residual                   HF:276-279, 338-341 */
#define XL_EPI_DGELU    3   /* C = acc * gelu_erf'(aux)          (backward of XL_EPI_GELU)            */
#define XL_EPI_TANH     4   /* C = tanh(acc+bias)                                        HF:566-572   */
#define XL_EPI_ROWMAX   5   /* no C: aux[(n/64)*M + m] = float4{max, sum exp(x - max), argmax (int bits), 0} of x = acc+bias over the
                             * 64-column segment n/64 of row m -- the sampler's softmax(-1).max(-1) over the 10k codebook without the
                             * logits ever reaching memory (ref tasks/imggen_model.py:229-235); finish with xl_rowmax_combine.  bf16
                             * operands, a_kmajor = b_kmajor = 1, M and N multiples of 256 (pad N with zero rows and bias -1e30) */
#define XL_EPI_GELU_DG  6   /* C = gelu_erf(acc+bias); aux = gelu_erf'(acc+bias): the derivative is saved instead of the
                             * pre-activation (same bytes; erf and exp(-x^2/2) are already in registers), so that the backward
                             * epilogue is a multiply -- XL_EPI_MULAUX -- instead of a second erf + exp per element        */
#define XL_EPI_MULAUX   7   /* C = acc * aux                      (backward of XL_EPI_GELU_DG)                            */

const char* xl_last_error(void);
int  xl_version(void);
/* Contexts: xl_ctx_create() -> handle (>= 1); xl_ctx_bind(handle) makes it the calling thread's current context (0 = the
 * default context; plan-able, so a recorded launch plan re-establishes its owner's context when replayed);
 * xl_ctx_destroy(handle) frees it (threads that still have it bound must bind another first).  One per host-side driver
 * object: the reference's counterpart is one nn.Module / one Trainer (ref lxmert_pretrain.py:47-108). */
int64_t xl_ctx_create(void);
int  xl_ctx_bind(int64_t ctx);
int  xl_ctx_destroy(int64_t ctx);
/* The setters below (xl_set_*, xl_gemm_trace, xl_gemm_set_workspace) write the BOUND context. */
/* 1: GEMM/SDPA transposed operands use ds_read_b64_tr_b16; 0: 16-bit LDS gathers (debug switch) */
int  xl_set_lds_transpose_read(int enable);
/* Step part of every dropout seed, in DEVICE memory: with a non-null pointer registered, each dropout site (xl_gemm
 * XL_EPI_RESIDUAL, xl_layernorm_bwd dx_dropped, xl_sdpa_fwd/bwd, xl_dropout) draws its mask from
 *     seed argument + 1000003 * *step_seed        (read by the kernel when it runs)
 * so a captured launch sequence (hipGraph replay of a whole training step) gets fresh masks after one 8-byte update of
 * *step_seed.  NULL (default): the seed argument alone.  The pointer is sampled when a launch is issued. */
int  xl_set_step_seed_ptr(const uint64_t* step_seed);
/* GEMM kernel choice for bf16 operands: 0 = 128x128 kernel only, 1 = by shape (default), 2 = the 256x256 ping-pong
 * kernel whenever the operands allow it (tuning / test switch; env XL_GEMM_PP sets the initial value) */
int  xl_set_gemm_pingpong(int mode);
/* 128x192 "duo" tiles of the ping-pong kernel: four waves and 80 KiB of LDS per workgroup, TWO workgroups per CU, so that one's
 * prologue / epilogue / hand-over runs under the other's K loop and workgroups of different streams can share a CU.  Eligible:
 * forward / dX layouts, M % 128 == 0, N % 192 == 0, bf16 output through a fast epilogue.  0 = never, 1 (default) = eligible
 * launches of at most XL_GEMM_DUO_MAX_TILES (64) tiles of 256x256 -- the language stream's 3328-row contractions, which otherwise
 * run on the 128x128 kernel at 0.10 MFMA-busy: -0.15..-0.25 ms per step --, 2 = whenever eligible (measured equal to the whole-CU
 * tiles on the large launches); env XL_GEMM_DUO.  Bit-identical results to the other tile shapes. */
int  xl_set_gemm_duo(int mode);
/* debug: when `buffer` is non-null (device memory, 4 x uint64 per workgroup of the largest launch), the ping-pong GEMM
 * kernel records wall-clock stamps (100 MHz) at start / after prologue / after the K loop / after its stores */
int  xl_gemm_trace(void* buffer);
/* debug / test: the row kernels' wave reductions run on the VALU's cross-lane paths (v_permlane32_swap, v_permlane16_swap, DPP) with
 * the pairing of the xor-shuffle butterfly; this evaluates both forms on n_waves x 64 inputs: sum_new / max_new (cross-lane form, as
 * used by LayerNorm / softmax / cross-entropy kernels) and sum_ref (the __shfl_xor form), each n_waves x 64 floats -- every lane's
 * result.  The test asserts sum_new == sum_ref bit for bit. */
int  xl_wave_reduce_check(const float* in, float* sum_new, float* max_new, float* sum_ref, int n_waves, void* stream);

/* ---------------------------------------------------------------- dense contractions (nn.Linear)
 * C[M,N] = alpha * sum_k A(m,k) * B(n,k)  [+ bias[n]]  -> epilogue
 *   a_kmajor: 1 -> A stored [M][K] (lda >= K);  0 -> A stored [K][M] (lda >= M)
 *   b_kmajor: 1 -> B stored [N][K] (ldb >= K);  0 -> B stored [K][N] (ldb >= N)
 *   forward  y = x W^T + b         (HF:232-234,277,326,339,469; ref lxrt/modeling.py:42,47): a_kmajor=1,b_kmajor=1
 *   backward dx = dy W                                                                      : a_kmajor=1,b_kmajor=0
 *   backward dW = dy^T x  (fp32 out, accumulate!=0 adds into C; split-K uses fp32 atomics)  : a_kmajor=0,b_kmajor=0
 *   in_dtype: element type of A,B,residual,aux.  out_dtype: element type of C (XL_F32 allowed with bf16 inputs).
 *   bias: fp32 [N] or NULL.  residual/aux: [M,N] with ldr/ldx or NULL.
 *   dropout (XL_EPI_RESIDUAL only): keep-prob (1-p_drop), mask = hash(seed, row m, column pair n>>1) -> one 16-bit draw per
 *     column (csrc/common.h dropout_draw16); p_drop=0 disables.
 *   colsum_out: NULL, or fp32 [N]: colsum_out[n] += sum_m C[m,n] over the values as stored in C (the bias gradient of the
 *     Linear layer whose output gradient C is, e.g. d(pre-activation) -> d(intermediate.dense.bias), HF:325-331); needs
 *     colsum_ws with xl_workspace_floats(N) floats.  Computed in the epilogue when every output tile is interior and
 *     aligned, by a separate pass over C otherwise -- same result either way.
 */
int xl_gemm(const void* A, const void* B, void* C, const float* bias,
            const void* residual, void* aux,
            int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldx,
            int a_kmajor, int b_kmajor, int in_dtype, int out_dtype,
            int epilogue, float alpha, int accumulate,
            float p_drop, uint64_t seed, float* colsum_out, float* colsum_ws, void* stream);


/* Grouped weight gradients: for i in [0, count), count <= 8:
 *     C_i[M_i, N_i] (fp32) += sum_k A_i[k, m] * B_i[k, n]        (dW = dY^T X; both operands stored [K_i rows][features])
 * i.e. xl_gemm(..., a_kmajor=0, b_kmajor=0, out fp32, accumulate=1) for several Linear layers in ONE launch: the output
 * tiles of all problems are dealt to the CUs together, so a K split of 2-3 fills the chip where a single d x d weight
 * needs 7-28 (each split is a pass of fp32 atomics over the output).  Replaces the per-parameter .grad accumulation of
 * autograd for the Linear weights of one LXMERT block (HF:258-342).  Arrays are HOST arrays of length count. */
int  xl_gemm_wgrad_group(const void* const* A, const void* const* B, void* const* C,
                         const int* M, const int* N, const int* K, const int* lda, const int* ldb, const int* ldc,
                         int count, int overwrite_mask, int dtype, void* stream);
/* the K split a grouped launch of these problems takes (1: every output tile has one writer -- what overwrite_mask wants: a split
 * launch honours the mask by clearing C first, which costs more than the optimizer pass's clear it replaces) */
int  xl_gemm_wgrad_group_splitk(const int* M, const int* N, const int* K, int count);
/* overwrite_mask: bit i set -> C_i = A_i^T B_i instead of C_i += ... (a weight with exactly ONE gradient contribution per
 * step: the training step then neither clears nor re-reads 4 bytes per parameter -- the clear moves out of xl_adamw, decay_flags
 * bit 2, and the epilogue's read-modify-write becomes a plain store).  The semantics hold whatever strategy the launch takes: a
 * problem whose tiles have one writer is stored plainly, one that is K-split (fp32 atomics) or has ragged tiles is cleared first. */

/* Slab workspace of the ping-pong kernel: per-stream, caller-owned device memory (16-byte aligned) in which the K slices
 * of one output tile meet -- every slice's workgroup writes its partial tile, the last to arrive sums them and runs the
 * epilogue.  With one registered for `stream`, launches on that stream use it for
 *   (a) the tail split: a launch of more than one round of tiles whose last round would be nearly empty (264 tiles on 256
 *       CUs) and whose contraction is deep runs those last tiles as K slices that fill the CUs as the previous round drains
 *       (the 10k-codebook contractions of the masked-row head: 446 -> ~250 us);
 *   (b) weight-gradient K splits (xl_gemm a_kmajor=b_kmajor=0 fp32 out, xl_gemm_wgrad_group) when
 *       xl_set_gemm_wgrad_slabs(1): one read-modify-write pass over C by the last arriver instead of a pass of fp32 atomics
 *       per split -- a fixed summation order (deterministic weight gradients), measured 1-3 % slower than the atomics.
 * xl_gemm_workspace_bytes(slabs): bytes for `slabs` partial tiles (256 covers every chip-filling launch; a launch that
 * needs more falls back).  ws = NULL unregisters.  A workspace must not be shared by streams that run concurrently. */
int64_t xl_gemm_workspace_bytes(int slabs);
int  xl_gemm_set_workspace(void* ws, int64_t bytes, void* stream);
int  xl_set_gemm_wgrad_slabs(int on);
/* tail split thresholds: at most `max_tail_tiles` tiles in the last round (0 disables; default 64) and a contraction of at
 * least `min_k` (default 4096: measured gain at K = 10000, none at K = 2048). */
int  xl_set_gemm_tail_split(int max_tail_tiles, int min_k);

/* ---------------------------------------------------------------- LayerNorm (eps inside sqrt, HF:188 et al.)
 * y = (x-mean)*rstd*gamma+beta over the last dim N; saves mean,rstd (fp32 [M]).  */
int xl_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, int M, int N, float eps, int dtype, void* stream);
/* dx from dy; dgamma/dbeta (fp32 [N]) are ACCUMULATED.  The LayerNorm input was `dropout(dense(.)) + residual`
 * (HF:277-279, 339-341): if dx_dropped != NULL and p_drop > 0 the kernel also writes dx_dropped = dx * mask(seed, m*N+n)
 * (the gradient entering the dense layer; same counter-based mask as the forward epilogue) .  If dbias_prev != NULL it
 * accumulates the column sums of that tensor (dx_dropped if written, else dx) = the dense layer's bias gradient. */
int xl_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                     const float* rstd, void* dx, float* dgamma, float* dbeta, float* dbias_prev,
                     int M, int N, float* workspace, void* dx_dropped, float p_drop, uint64_t seed,
                     int dtype, void* stream);
/* Column reductions (LayerNorm affine / bias gradients, column sums) run as a two-stage reduction through a
 * caller-owned fp32 `workspace` of at least xl_workspace_floats(N) elements (per-block partial slabs + one
 * combine launch); workspace == NULL falls back to fp32 atomics on the output. */
int64_t xl_workspace_floats(int N);

/* visual feature encoder tail (HF:468-476): y = (LN_v(xv) + LN_b(pos W_b^T + b_b)) / 2
 * xv = visn_fc output [M,N]; pos fp32 [M,P] (P<=8); box weights fp32 [N,P].
 * saves mean/rstd of both LayerNorms and the box pre-LN row is recomputed in backward. */
int xl_visn_ln_fwd(const void* xv, const float* pos, const float* wbox, const float* bbox,
                   const float* gv, const float* bv, const float* gb, const float* bb,
                   void* y, float* mean_v, float* rstd_v, float* mean_b, float* rstd_b,
                   int M, int N, int P, float eps, int dtype, void* stream);
int xl_visn_ln_bwd(const void* dy, const void* xv, const float* pos, const float* wbox, const float* bbox,
                   const float* gv, const float* gb,
                   const float* mean_v, const float* rstd_v, const float* mean_b, const float* rstd_b,
                   void* dxv, float* dgv, float* dbv, float* dgb, float* dbb,
                   float* dwbox, float* dbbox, float* dbias_visn,
                   int M, int N, int P, float* workspace, int dtype, void* stream);

/* The second stage of the two-stage column reductions (LayerNorm affine / bias gradients: xl_layernorm_bwd, xl_visn_ln_bwd,
 * xl_colsum, xl_masked_colsum, xl_gemm's colsum_out) can be deferred: with xl_set_deferred_reduce(1) those calls only write
 * their partial slabs and remember what is left to do, and xl_flush_reductions(stream) combines everything pending on that
 * stream in one launch.  While deferred, every such call needs a workspace region of its own. */
int xl_set_deferred_reduce(int on);
int xl_flush_reductions(void* stream);
/* the same combine launch for everything pending from `producer_stream`, issued on `launch_stream` (which the caller has ordered
 * after the producers: xl_stream_fork): the training step rides its per-layer combines on the weight-gradient companion stream, off
 * the dX dependency chain (~25 six-microsecond launches and their launch gaps per step) */
int xl_flush_reductions_on(void* producer_stream, void* launch_stream);

/* ---------------------------------------------------------------- embeddings (HF:191-214)
 * y[b,l] = LN(word[ids[b,l]] + pos[l] + type[tt[b,l]]); tables in `dtype`; saves pre-LN sum + stats. */
int xl_embed_ln_fwd(const int64_t* ids, const int64_t* tt, const void* word, const void* pos,
                    const void* type, const float* gamma, const float* beta,
                    void* y, void* pre, float* mean, float* rstd,
                    int B, int L, int N, float eps, int dtype, void* stream);
/* scatter-add of d(pre) into the fp32 table gradients.  Row 0 of every table is frozen
 * (nn.Embedding(padding_idx=0), HF:184-186).  Deterministic: every table row has ONE writer, which adds the rows that map to it
 * in row order (no floating-point atomics).
 *   order: int32 [B*L], the row indices b*L+l sorted by (ids[row], row) -- a stable argsort of the flattened ids, computed where the
 *          ids are made (the data loader, next to attention_mask / masked_rows); NULL: every row scans the batch for its id
 *          (same result, ~40x slower at B*L = 5120).
 *   tt / n_types: token-type ids and type_vocab_size (HF: 2); tt NULL = every token has type 0 (no gradient: frozen row). */
int xl_embed_bwd(const void* dpre, const int64_t* ids, const int64_t* tt, const int32_t* order,
                 float* dword, float* dpos, float* dtype_tab, int B, int L, int N, int n_types, int dtype, void* stream);

/* ---------------------------------------------------------------- codebook input (ref lxrt/modeling.py:185-193)
 * feats[b,v,:] = vis_mask[b,v] ? mask_feat : centroids[cluster_ids[b,v]]   (centroids in `dtype`) */
int xl_codebook_gather(const int64_t* cluster_ids, const uint8_t* vis_mask, const void* centroids,
                       const float* mask_feat, void* feats, int M, int F, int dtype, void* stream);
/* out[n] += sum over rows m with mask[m]!=0 of x[m,n]    (d mask_feat pre-image; also generic masked colsum) */
int xl_masked_colsum(const void* x, const uint8_t* mask, float* out, int M, int N, int ldx, float* workspace,
                     int dtype, void* stream);
/* out[n] += sum_m x[m,n]   (bias gradients) */
int xl_colsum(const void* x, float* out, int M, int N, int ldx, float* workspace, int dtype, void* stream);

/* y[m,n] = x[m,n] * (keep(seed, m, n) ? 1/(1-p) : 0) (rows of ldx / ldy elements): the
 * counter-based mask used by every dropout site of the path (HF:213,258,278,340,475); calling it again with the same
 * (seed, p) on a gradient applies the identical mask (backward).  In-place (y == x) allowed. */
int xl_dropout(const void* x, void* y, int M, int N, int ldx, int ldy, float p_drop, uint64_t seed, int dtype, void* stream);

/* dx = dy * gelu_erf'(pre), n elements (head transform backward, HF:582-586) */
int xl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream);

/* dx = dy * (1 - y*y), n elements: backward of LxmertPooler's tanh (HF:566-572) given its output y */
int xl_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);

/* BCEWithLogitsLoss(reduction='mean') of the VQA/GQA fine-tune step (ref tasks/vqa.py:73,187; gqa.py:70,150) over fp32
 * logits [M,N] and soft targets [M,N], plus its gradient:  loss[0] += mean(max(x,0) - x t + log(1+exp(-|x|)));
 * dlogits (element type `dtype`, row stride ld_dlogits >= N, pad columns written as 0; NULL = loss only)
 *   = (sigmoid(x) - t) / (M N).  `loss` must be zeroed by the caller. */
int xl_bce_logits_fwd_bwd(const float* logits, const float* targets, void* dlogits, float* loss,
                          int M, int N, int ld_logits, int ld_targets, int ld_dlogits, int dtype, void* stream);

/* Iterative (Mask-Predict) sampler, ref tasks/imggen_model.py:169-257, kept on the device between steps:
 * xl_remask_lowest: vis_mask[b, v] (uint8) = 1 at the n_mask lowest-confidence positions of row b (prob fp32 [B,V], V <= 64;
 *   ties -> lower index), 0 elsewhere  (ref :204-212  topk(largest=False) + scatter_).
 * xl_sampler_update: code_ids[i] (int64) = pred_ids[i] (int32, from xl_ce_fwd_bwd's argmax) where vis_mask[i] != 0
 *   (ref :238-243; the [B,V,2048] code tensor is represented by its codebook ids + mask, materialised by
 *   xl_codebook_gather exactly as `where(mask, mask_feat, vis_emb(ids))`). */
int xl_remask_lowest(const float* prob, void* vis_mask, int B, int V, int n_mask, void* stream);
int xl_sampler_update(const int* pred_ids, const void* vis_mask, int64_t* code_ids, int n, void* stream);
/* Autoregressive variant (ref tasks/imggen_model.py:49-167): per step ONE position per image takes its prediction and is
 * un-masked: fixed_pos >= 0 -> that position for every image (position_TLBR / position_random order drawn on the host),
 * fixed_pos < 0 -> each image's most confident position among the not yet visited ones (position_confidence, ref :140-149;
 * `visited` uint8 [B,V] is updated). */
int xl_sampler_ar_update(const float* prob, const int* pred_ids, void* visited, void* vis_mask, int64_t* code_ids,
                         int B, int V, int fixed_pos, void* stream);

/* ---------------------------------------------------------------- attention core (HF:247-263)
 * per (b,h): O = softmax(Q K^T * scale, keys with key_mask==0 excluded) V ; nq, nk <= 64 on the on-chip kernels (the path's
 * shapes: 20 text tokens, 8x8 grid); 65..512 (--max_text_length up to the position table, ref param.py:140) on plain long-sequence
 * kernels (fp32 arithmetic, one lane per query / key; xl_sdpa_bwd then needs `workspace`).
 * q/k/v/o are [B, n, H*dh]-shaped views with row strides ldq/ldk/ldv/ldo (elements); head h
 * occupies columns [h*dh, (h+1)*dh).  key_mask: uint8 [B,nk] or NULL.  lse: fp32 [B,H,nq]
 * (log-sum-exp of the scaled scores) saved for backward.  Probability dropout (HF:258):
 * p_drop with mask hash(seed, row = (b*H+h)*nq+q, column = key).
 * PACKED rows (q_rowoff / k_rowoff, int32 device arrays [B+1], ascending; NULL = the dense [B, n] layout): the rows of batch
 * element b on that side are [off[b], off[b+1]) of the matrix -- the language rows of a batch with the [PAD] positions removed
 * (the additive mask of HF:238-266 excludes exactly those keys, and no loss reads a [PAD] query's output: dropping the rows is
 * exact).  nq / nk stay the per-example CAPACITY (<= 64; lse and the dropout counters keep their [B, H, n] indexing); a packed
 * key side needs no key_mask.  Rows [off[B], *_rows_padded) -- the tail that rounds the packed row count up to the GEMM row
 * tile -- are written as ZEROS (o; dq on the query side; dk, dv on the key side), so the contractions over all rows that
 * follow (out-projection, weight gradients) read zeros there.
 * keep_bits (optional; NULL = the backward evaluates the mask hash again): xl_sdpa_keep_bits_bytes(...) bytes, 16-byte aligned,
 * in which the forward leaves its dropout decisions (one bit per (query, key) of every (b, h) problem, in the kernel's register
 * order) for xl_sdpa_bwd of the same call geometry: what autograd keeps as the dropout mask of HF:258, at 1/16 of its bytes.
 * Same mask, same results bit for bit; the backward's two passes then test a bit instead of hashing (the hash was a quarter of
 * its instructions).  Only on the on-chip bf16 kernels: xl_sdpa_keep_bits_bytes returns 0 for any other geometry, and passing
 * a buffer where the kernels would not use it (long sequences, fp32, operands that are not 16-byte aligned) is XL_ERR_BAD_ARG. */
int64_t xl_sdpa_keep_bits_bytes(int B, int H, int nq, int nk, int dh, int dtype);
int xl_sdpa_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask,
                void* o, float* lse, int B, int H, int nq, int nk, int dh,
                int ldq, int ldk, int ldv, int ldo, float scale,
                float p_drop, uint64_t seed, const int* q_rowoff, const int* k_rowoff, int q_rows_padded, int k_rows_padded,
                uint32_t* keep_bits, int dtype, void* stream);
int xl_sdpa_bwd(const void* q, const void* k, const void* v, const uint8_t* key_mask,
                const void* dout, const float* lse,
                void* dq, void* dk, void* dv, int B, int H, int nq, int nk, int dh,
                int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv, float scale,
                float p_drop, uint64_t seed, float* bias_grad, float* workspace,
                const int* q_rowoff, const int* k_rowoff, int q_rows_padded, int k_rows_padded,
                const uint32_t* keep_bits, int dtype, void* stream);
/* Attention probabilities of one attention block, for LxmertModel.forward(output_attentions=True) (HF:691-704, 238-266: the
 * softmax AFTER its dropout): probs fp32 [B, H, nq, nk] (dense, also for packed rows: queries / keys beyond an example's length
 * and masked keys give zeros), recomputed from q, k and the lse that xl_sdpa_fwd saved -- the fused forward never stores them. */
int xl_attn_probs(const void* q, const void* k, const uint8_t* key_mask, const float* lse, float* probs,
                  int B, int H, int nq, int nk, int dh, int ldq, int ldk, float scale, float p_drop, uint64_t seed,
                  const int* q_rowoff, const int* k_rowoff, int dtype, void* stream);
/* bias_grad (optional, fp32 [3*H*dh]): bias_grad[q | k | v] += column sums of dq / dk / dv over all B*n rows - the bias
 * gradients of the query/key/value projections (HF:232-239 nn.Linear) - from per-token scalars inside the kernel, without
 * re-reading dq/dk/dv; `workspace` as for xl_colsum (the second stage obeys xl_set_deferred_reduce). */

/* ---------------------------------------------------------------- head losses (ref lxrt/modeling.py:237-290)
 * counts[0] = #labels != -100 ; nmask[b] = sum_v vis_mask[b,v]            (device-side, no host sync) */
int xl_mask_counts(const int64_t* labels, const uint8_t* vis_mask, float* counts, float* nmask,
                   int B, int V, void* stream);
/* CrossEntropyLoss(ignore_index=-100, mean): logits fp32 [M,K] (ldl); loss_out[0] += sum_rows nll / count;
 * dlogits (`dtype`, lddl) = (softmax - onehot) * grad_scale / count for valid rows, 0 otherwise.
 * Also writes per-row lse / argmax (fp32 / int32, may be NULL) for the sampler. */
int xl_ce_fwd_bwd(const float* logits, const int64_t* labels, const float* counts,
                  void* dlogits, float* loss_out, float* row_lse, int32_t* row_argmax, float* row_maxprob,
                  int M, int K, int ldl, int lddl, float grad_scale, int dtype, void* stream);
/* masked SmoothL1 feature regression (ref lxrt/modeling.py:273-287): target row m = targets[m,:] when `targets` != NULL
 * (label_dict['feat_labels'], [B*V, F] in `dtype`: the real grid features of lxmert_pretrain.py:177-179), else
 * centroids[cluster_ids[m]] (never masked);
 * loss_out[0] += mean_b( sum_v mask*mean_F sl1 / max(nmask_b,1) ); dpred written for every row. */
int xl_featloss_fwd_bwd(const void* pred, const void* centroids, const int64_t* cluster_ids,
                        const uint8_t* vis_mask, const float* nmask, void* dpred, float* loss_out,
                        int B, int V, int F, float grad_scale, const int* rows, int n_rows, const void* targets,
                        int dtype, void* stream);
/* Both head losses only read the masked positions (labels are -100 / the SmoothL1 term is multiplied by vis_mask elsewhere:
 * ref lxrt/modeling.py:253-256, 273-287), so the training step runs the head on the masked rows only:
 * xl_gather_rows: dst[r,:] = src[rows[r],:]; xl_scatter_rows: dst[rows[r],:] = src[r,:]  (rows: int32 [n_rows], ascending row
 * ids b*V+v of the masked positions; N, ld multiples of the 16-byte vector).  xl_featloss_fwd_bwd with rows != NULL takes
 * pred / dpred with one row per entry of `rows`. */
int xl_gather_rows(const void* src, const int* rows, void* dst, int n_rows, int N, int ld_src, int ld_dst, int dtype, void* stream);
int xl_scatter_rows(const void* src, const int* rows, void* dst, int n_rows, int N, int ld_src, int ld_dst, int dtype, void* stream);
/* A row list may be PADDED with negative entries (the caller rounds its length up so that launch sizes do not depend on the
 * batch: a captured step replays unchanged): xl_gather_rows writes a zero row for them, xl_scatter_rows skips them,
 * xl_featloss_fwd_bwd gives them no loss and a zero gradient row, and xl_gather_labels hands the cross-entropy the ignore
 * label:  out[r] = rows[r] >= 0 ? labels[rows[r]] : -100. */
int xl_gather_labels(const int64_t* labels, const int* rows, int64_t* out, int n_rows, void* stream);
/* Second half of XL_EPI_ROWMAX: per row m, over the n_seg segment records ws[seg*M + m] of the GEMM epilogue:
 * row_argmax[m] = argmax_n x (lowest index on ties, as torch.max), row_maxprob[m] = softmax(x)[argmax] = 1 / sum_n exp(x_n - max),
 * row_lse[m] = max + log(sum) (any output may be NULL). */
int xl_rowmax_combine(const float* ws, int n_seg, int M, float* row_maxprob, int* row_argmax, float* row_lse, void* stream);

/* ---------------------------------------------------------------- optimizer side (ref lxmert_pretrain.py:343-364)
 * sumsq[0] += sum g^2 over n fp32 elements.  Deterministic: block partials are added in a fixed order by the last block to
 * arrive, so every rank of a data-parallel job derives the same clip factor from the same reduced gradients (an atomic per block
 * left replicas 1 ulp apart after one step).  One plain update of sumsq[0] per call: do not run two calls on the same sumsq
 * concurrently.  `scratch`: caller-owned device memory of xl_sumsq_scratch_bytes() bytes (16-byte aligned), ZERO before its
 * first use and left zero-ticketed by every call; it holds the block partials and the arrival ticket, so two calls that may
 * run concurrently (different streams, different devices) need one each. */
int64_t xl_sumsq_scratch_bytes(void);
int xl_sumsq(const float* g, float* sumsq, int64_t n, void* scratch, void* stream);
/* Device-side update counter and schedule (ref lxmert_pretrain.py:138-139 get_linear_schedule_with_warmup; 4.1.1 AdamW bias
 * corrections): *step += 1 (t = the update about to be applied), lr_and_steps = {base_lr * schedule(t-1), 1-beta1^t,
 * 1-beta2^t, t}.  Stream-ordered before xl_adamw, so the host never writes step scalars into memory a queued step reads. */
int xl_schedule_step(int64_t* step, float base_lr, int warmup_steps, int total_steps, float beta1, float beta2,
                     float* lr_and_steps, void* stream);
/* transformers==4.1.1 AdamW on flat fp32 buffers with fused gradient clipping:
 * clip = min(1, max_norm/(sqrt(sumsq[0])+1e-6)) (max_norm<=0 disables), g' = g*clip*grad_scale;
 * decay_flags: uint8 per 256-element chunk: bit 0 = apply weight decay, bit 2 = with zero_grad, do NOT clear this chunk's gradient
 * (the next backward overwrites it: xl_gemm_wgrad_group overwrite_mask), bit 1 = SKIP the chunk (a tensor that got no
 * gradient this step: the reference resets .grad to None every step and AdamW skips such tensors -- the task round-robin
 * of lxmert_pretrain.py:296-298 changes the set every step).  lr_and_steps (device, fp32[4]):
 * {lr, bias_corr1 = 1-beta1^t, bias_corr2 = 1-beta2^t, unused}; chunk_steps (device int32 per chunk, may be NULL): the
 * per-tensor update count state["step"] of transformers' AdamW -- when given, the bias corrections are computed from it
 * instead of lr_and_steps[1..2] (the caller increments it for the chunks it does not skip).  Writes the compute copy
 * (`dtype`) of every updated parameter to p_compute (may be NULL).  zero_grad != 0: the gradient of every updated chunk is
 * cleared in the same pass (the trainer's optimizer.zero_grad(), ref lxmert_pretrain.py:336 -- 4 bytes per element written
 * here instead of a separate 0.8 GB clear before the next backward). */
int xl_adamw(float* p, float* g, float* m, float* v, void* p_compute,
             const uint8_t* decay_flags, const int* chunk_steps, const float* sumsq, const float* lr_and_steps,
             int64_t n, float beta1, float beta2, float eps, float weight_decay, float max_norm,
             float grad_scale, int zero_grad, int dtype, void* stream);
/* dst (`dtype`) = src (fp32), n elements */
int xl_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, void* stream);
/* dst (fp32) = src (`dtype`) */
int xl_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream);
/* Sparse fp32 side car of the sharded exchange (trainer collective="rs+ag", gather="bf16": SURVEY 5.8 "bf16 on the wire").  idx: int32
 * [n] positions (relative to src / dst) of the elements that are read in fp32 (biases, LayerNorm affines ...):
 *   xl_take_f32: dst[j] = src[idx[j]] if own_lo <= idx[j] < own_hi else 0   (the pack a rank contributes to a sum all-reduce)
 *   xl_put_f32:  dst[idx[j]] = src[j]                                        (the all-reduced pack back into the master buffer) */
int xl_take_f32(const float* src, const int32_t* idx, int n, int own_lo, int own_hi, float* dst, void* stream);
int xl_put_f32(float* dst, const int32_t* idx, int n, const float* src, void* stream);

/* ---------------------------------------------------------------- launch plans (csrc/plan.hip)
 * The host language records one training step as the list of C-ABI calls it made (function + argument words) and replays
 * it with ONE call per step: same kernels, streams and events as the eager sequence, no interpreter in between.  What a
 * step varies must live in device memory: inputs (caller's static buffers), the dropout step seed (xl_set_step_seed_ptr),
 * schedule scalars (xl_schedule_step), row lists padded to a fixed length (xl_gather_labels).
 *   xl_memset / xl_stream_fork: the two stream operations a step needs besides kernels, as plan-able calls
 *     (xl_stream_fork = hipEventRecord(event, from) + hipStreamWaitEvent(to, event): `to` continues after everything queued
 *     on `from` so far; events come from xl_event_create, one per fork site is enough).
 *   xl_plan_fn_id(name): index of a plan-able entry point (negative: it cannot be part of a plan); xl_plan_fn_nargs.
 *   xl_plan_create(n_calls, fn_ids, n_args, words): words = the calls' arguments back to back, one 64-bit word each
 *     (pointers / integers by value, floats as their 32-bit pattern; host arrays, as xl_gemm_wgrad_group takes them, must
 *     outlive the plan).  Returns a handle (0 on error).  xl_plan_run(handle) replays; xl_plan_destroy frees. */
int  xl_memset(void* dst, int value, int64_t bytes, void* stream);
int64_t xl_event_create(void);
int  xl_event_destroy(int64_t event);
int  xl_stream_fork(void* event, void* from_stream, void* to_stream);
/* the two halves of xl_stream_fork on their own (plan-able): a consumer stream waits, at a later point of ITS queue, for a point
 * recorded earlier on a producer stream -- the next step's forward waiting layer by layer for the optimizer pass of the step before */
int  xl_event_record(void* event, void* stream);
int  xl_stream_wait(void* event, void* stream);
int  xl_plan_fn_id(const char* name);
int  xl_plan_fn_nargs(int fn_id);
int64_t xl_plan_create(int n_calls, const int* fn_ids, const int* n_args, const uint64_t* words);
int  xl_plan_run(int64_t plan);
int  xl_plan_destroy(int64_t plan);

/* ---------------------------------------------------------------- gradient exchange (RCCL over xGMI; csrc/comm.hip)
 * The data-parallel step of the reference is DistributedDataParallel around the model (ref pretrain/lxmert_pretrain.py:102-106,
 * 694-700: one process per GPU, NCCL backend).  These calls put the same collectives behind the C ABI so that a launch plan can
 * contain them: one communicator per process, every collective on the communicator's OWN stream, ordered after everything
 * queued so far on `after_stream` (event record + wait, issued by the call), all asynchronous.
 *   xl_comm_unique_id(id128): rank 0 fills a 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *     (xlxmert_amd.trainer broadcasts it through torch.distributed);  xl_comm_init(id128, rank, nranks, comm_stream) -> handle
 *     (0 on error: collective over all ranks, each with its GPU current).  comm_stream: the caller's stream for the collectives
 *     (NULL: the library creates one) -- HIP binds a stream to one of its few hardware queues at FIRST USE, and a collective
 *     stream that lands on a compute stream's queue blocks that stream's kernels behind its event waits (+2.6 ms per step
 *     measured): create and first-use it together with the compute streams (engine.reserve_streams(comm=True)).  xl_comm_destroy.
 *   xl_comm_allreduce: buf (count elements of `dtype`, XL_F32 or XL_BF16) := sum over ranks, in place.
 *   xl_comm_reduce_scatter / xl_comm_allgather: the two halves (recv_count / send_count = elements per rank): rank r ends up
 *     with the sum of everybody's r-th piece / everybody ends up with every rank's piece.
 *   xl_comm_bcast(buf, bytes, root): parameter / optimizer-state broadcast at start-up (DDP's constructor broadcast);
 *   xl_comm_reduce: sum onto `root` (the epoch metrics of ref utils.py:11-39).
 *   xl_comm_wait(comm, stream): `stream` continues after every collective issued so far.
 * RCCL is bound at run time (dlopen librccl.so.1): a process that never calls xl_comm_* does not need it. */
int  xl_comm_unique_id(void* id128);
int64_t xl_comm_init(const void* id128, int rank, int nranks, void* comm_stream);
int  xl_comm_destroy(int64_t comm);
/* ranks of the communicator as RCCL reports them (ncclCommCount); -1: stale handle */
int  xl_comm_nranks(int64_t comm);
int  xl_comm_allreduce(int64_t comm, void* buf, int64_t count, int dtype, void* after_stream);
int  xl_comm_reduce_scatter(int64_t comm, const void* send, void* recv, int64_t recv_count, int dtype, void* after_stream);
int  xl_comm_allgather(int64_t comm, const void* send, void* recv, int64_t send_count, int dtype, void* after_stream);
int  xl_comm_bcast(int64_t comm, void* buf, int64_t bytes, int root, void* after_stream);
int  xl_comm_reduce(int64_t comm, void* buf, int64_t count, int dtype, int root, void* after_stream);
int  xl_comm_wait(int64_t comm, void* stream);


/* ------------------------------------------------------------------------------------------------------------------------------
 * EXPERIMENTAL entry points: kernel / schedule variants that were built, proven bit-exact and MEASURED SLOWER than the defaults
 * inside the four-stream training step (DESIGN.md section 6 has each one's numbers).  They are compiled and exported only by the
 * experimental build (XL_EXPERIMENTAL=1 python -m xlxmert_amd.build -> libxlxmert_hip_exp.so, -DXL_EXPERIMENTAL; tests that
 * exercise them are collected only with XL_EXPERIMENTAL=1); the default library neither contains their kernels nor exports these
 * symbols.
 * ------------------------------------------------------------------------------------------------------------------------------ */
#ifdef XL_EXPERIMENTAL
/* 256x192 output tiles of the ping-pong kernel (forward / dX layouts, N a multiple of 192, M of 256, bf16 output): 0 = never
 * (default since round 4: in the four-stream step a main-chain launch that leaves a quarter of the CUs to the other streams is worth
 * more than the full round it forgoes, -0.27 .. -0.37 ms per step), 1 = when they shorten the launch taken alone (N = 768 gives 192
 * tiles of 256x256 on 256 CUs but 256 of 256x192), 2 = whenever eligible (test switch); env XL_GEMM_BN192 sets the initial value */
int  xl_set_gemm_tile192(int mode);
/* persistent variant of the ping-pong kernel for launches of several rounds of 256x256 tiles with a short contraction (the FFN's
 * first Linear and the gradient through its GELU: N = 3072, K = 768): one workgroup per CU walks its tiles and requests the next
 * tile's first K tile under the epilogue of the current one.  0 = never (default; env XL_GEMM_PERSIST), 1 = when eligible.
 * Bit-identical results either way.  Measured: -5 % on the FFN1 + GELU launch in isolation, +0.1 ms on the whole step -- a
 * workgroup that holds its CU across tiles keeps the other streams' workgroups out at the tile boundaries. */
int  xl_set_gemm_persistent(int on);
/* 128x192 output tiles by EIGHT waves of 32x96 at 128 registers (csrc/gemm_q.hip): two workgroups per CU, four waves per SIMD -- each
 * workgroup keeps two waves per SIMD in its K loop, so one's prologue / epilogue / hand-over runs under the other's K loop (the whole-CU
 * 256x256 tile spends 30-40 % of a K = 768 tile's time there with nothing to overlap it).  0 = never, 1 = contractions of depth <=
 * XL_GEMM_Q_MAX_K (1024) and width N <= XL_GEMM_Q_MAX_N (2304) with between XL_GEMM_Q_MIN_TILES (64) and XL_GEMM_Q_MAX_TILES (unbounded) tiles of 128x192, 2 = every eligible launch (forward / dX layouts, bf16 in / out,
 * a fast epilogue kind, no fused column sums, M % 128 == N % 192 == K % 64 == 0); env XL_GEMM_Q sets the initial value.  Bit-identical
 * results to the other tile shapes. */
int  xl_set_gemm_q(int mode);
/* "relay" kernel (csrc/gemm_relay.hip): one persistent 8-wave workgroup per CU whose two groups of four waves trade roles every
 * 256x128 output tile -- one group runs the tile's K loop as a self-pipelined MFMA stream, the other issues its LDS-DMA and runs the
 * PREVIOUS tile's epilogue under it (two accumulator sets per SIMD, one per wave).  0 = never, 1 = launches of more than
 * XL_GEMM_RELAY_MIN_TILES (257) tiles of 256x256 with K <= XL_GEMM_RELAY_MAX_K (1024), 2 = every eligible launch (forward / dX layouts,
 * bf16 in / out, fast epilogue kind NONE / RESIDUAL / GELU_DG / MULAUX, no fused column sums, M % 256 == N % 256 == K % 64 == 0,
 * K >= 768); env XL_GEMM_RELAY sets the initial value.  Bit-identical results to the other tile shapes. */
int  xl_set_gemm_relay(int mode);
/* number of persistent workgroups a relay launch may put up (default 256 = one per CU; at most half the launch's 256x128 tiles);
 * tuning / test switch, env XL_GEMM_RELAY_WGS */
int  xl_set_gemm_relay_wgs(int wgs);
/* K split of a launch WITH an epilogue (forward / dX layouts, bf16 in and out, fast epilogue): a launch of at most
 * XL_GEMM_SPLIT_EPI_MAX_TILES (80) output tiles of 256x192 / 256x256 whose contraction is at least XL_GEMM_SPLIT_EPI_MIN_K (1536)
 * deep runs every tile as 2..4 K slices of >= 12 K tiles on whole-CU workgroups; the slices meet in the stream's slab workspace
 * (xl_gemm_set_workspace: needed), are summed in slice order by the last arriver (deterministic), which runs the epilogue.  The
 * language stream's 3328 packed rows against the d x dff and d x 3d weights are the case: 56 tiles, K = 3072 / 2304 -- measured 47 ->
 * 50 us at every split factor (the slab hand-over costs what the shorter K loop saves), hence OFF by default.
 * 0 = never (default; env XL_GEMM_SPLIT_EPI), 1 = when eligible (needs the stream's slab workspace, xl_gemm_set_workspace; without
 * one the launch runs unsplit).  Same values as the unsplit launch up to the fp32 summation order. */
int  xl_set_gemm_split_epi(int on);
/* Two contractions of one shape class, possibly in ONE launch: exactly
 *     xl_gemm(A0, B0, C0, bias0, residual0, aux0, M0, ..., accumulate 0, p_drop, seed0, colsum_out0, colsum_ws0, stream);
 *     xl_gemm(A1, B1, C1, bias1, residual1, aux1, M1, ..., accumulate 0, p_drop, seed1, colsum_out1, colsum_ws1, stream);
 * (same N, K, leading dimensions, layouts, element types, epilogue kind, alpha and dropout probability; own operands, row count,
 * dropout seed and column-sum outputs).  The visual and the language side of a cross-modality layer's self-attention / FFN
 * sub-blocks (HF:417-449: visn_self_att | lang_self_att, visn_inter/output | lang_inter/output) and of the two single-modality
 * stacks (HF:516-529) are such pairs: 16384 visual rows and ~3300 packed language rows against different weights of the same
 * shape.  When both problems are bf16 with K-major A, M a multiple of 256, N of 256, aligned operands and an epilogue kind with a
 * paired instance (forward layout: NONE / RESIDUAL / GELU_DG; dX layout: NONE / RESIDUAL / MULAUX), their 256x256 output tiles are
 * dealt to the CUs by one launch of the ping-pong kernel -- the language side's 39 row tiles ride in the CUs the visual side's
 * last round leaves idle instead of occupying a quarter of the chip at a tenth of its matrix rate beside it.  Otherwise: the two
 * xl_gemm calls.  Results are bit-identical either way (the same tile code runs each tile).  xl_set_gemm_pair(0) / env
 * XL_GEMM_PAIR=0: always two launches. */
int  xl_gemm_pair(const void* A0, const void* B0, void* C0, const float* bias0, const void* residual0, void* aux0, int M0,
                  uint64_t seed0, float* colsum_out0, float* colsum_ws0,
                  const void* A1, const void* B1, void* C1, const float* bias1, const void* residual1, void* aux1, int M1,
                  uint64_t seed1, float* colsum_out1, float* colsum_ws1,
                  int N, int K, int lda, int ldb, int ldc, int ldr, int ldx, int a_kmajor, int b_kmajor, int in_dtype,
                  int out_dtype, int epilogue, float alpha, float p_drop, void* stream);
int  xl_set_gemm_pair(int on);
#endif /* XL_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif
