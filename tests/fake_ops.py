"""TEST INFRASTRUCTURE: a host-memory stand-in for xlxmert_amd.ops.HipOps with the same method signatures,
written with plain torch CPU math (the argument conventions of include/xlxmert_hip.h, nothing else).

It exists so that the engine's kernel *sequencing* (forward chain, hand-derived backward chain, buffer/stride
bookkeeping, gradient placement) can be checked against the oracle in the GPU-less build container.  It is never
importable from the product package and is not a fallback: `xlxmert_amd` only ever constructs HipOps.
"""
import math

import torch

EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_TANH, EPI_ROWMAX, EPI_GELU_DG, EPI_MULAUX = 0, 1, 2, 3, 4, 5, 6, 7


def v2(t, rows, cols, ld):
    return torch.as_strided(t, (rows, cols), (ld, 1))


def gelu_grad(x):
    cdf = 0.5 * (1.0 + torch.erf(x * 0.7071067811865476))
    pdf = 0.3989422804014327 * torch.exp(-0.5 * x * x)
    return cdf + x * pdf


def keep_scale(seed, row, col, p_drop):
    """Host restatement of csrc/common.h dropout_draw16()/dropout_scale(): one 32-bit multiply-xorshift mix of
    (row, col >> 1, seed) gives two 16-bit draws (even / odd column); keep iff draw >= p * 2^16.
    row, col: broadcastable int64 tensors.  Returns float32 0 or 1/(1-p)."""
    import numpy as np
    u32 = np.uint32
    row, col = torch.broadcast_tensors(torch.as_tensor(row), torch.as_tensor(col))
    r = (row.numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(u32)
    c = (col.numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(u32)
    s_lo, s_hi = u32(seed & 0xFFFFFFFF), u32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        mix = s_lo ^ (s_hi * u32(0xC2B2AE3D))
        h = r * u32(0x9E3779B1) ^ (c >> u32(1)) * u32(0x85EBCA77) ^ mix
        h ^= h >> u32(16); h = h * u32(0x7FEB352D)
        h ^= h >> u32(15); h = h * u32(0x846CA68B)
        h ^= h >> u32(16)
    draw = np.where((c & u32(1)) != 0, h >> u32(16), h & u32(0xFFFF))
    thr = u32(int(np.float32(p_drop) * np.float32(65536.0)))
    keep = (draw >= thr).astype(np.float32) * np.float32(1.0 / (1.0 - p_drop))
    return torch.from_numpy(keep)


def dropout_keep_matrix(seed, n_problems, nq_cap, n_rows, n_cols, p_drop):
    """bool [n_problems, n_rows, n_cols]: the attention kernels' keep decision for (row = problem * nq_cap + q, column = key) --
    also for the q >= nq_cap / key >= nk positions that a 32-wide fragment carries along (the kernels hash them like any other)"""
    row = (torch.arange(n_problems).view(-1, 1, 1) * nq_cap + torch.arange(n_rows).view(1, -1, 1))
    col = torch.arange(n_cols).view(1, 1, -1)
    return keep_scale(seed, row, col, p_drop) > 0


class FakeOps:
    def __init__(self, dtype):
        self.dtype = dtype
        self.calls = []
        self.step_seed = None

    def zero(self, t):
        t.zero_()

    def stream_fork(self, from_stream, to_stream):
        pass

    def set_step_seed_ptr(self, step_seed):
        """xl_set_step_seed_ptr: the kernels add 1000003 * *step_seed to every dropout seed while the pointer is set"""
        self.step_seed = step_seed

    def _seed(self, seed):
        return seed if self.step_seed is None else seed + int(self.step_seed.item()) * 1000003

    def set_lds_transpose_read(self, enable):
        pass

    def gemm(self, A, B, C, bias, residual, aux, M, N, K, lda, ldb, ldc, ldr=0, ldx=0, a_kmajor=1, b_kmajor=1,
             out_f32=False, epilogue=EPI_NONE, alpha=1.0, accumulate=0, p_drop=0.0, seed=0, colsum=None, ws=None):
        self.calls.append(("gemm", M, N, K, a_kmajor, b_kmajor, epilogue))
        a = (v2(A, M, K, lda) if a_kmajor else v2(A, K, M, lda).t()).float()
        b = (v2(B, N, K, ldb) if b_kmajor else v2(B, K, N, ldb).t()).float()
        acc = alpha * (a @ b.t())
        if bias is not None:
            acc = acc + torch.as_strided(bias, (N,), (1,)).float()[None, :]
        if epilogue == EPI_GELU:
            v2(aux, M, N, ldx).copy_(acc)
            acc = torch.nn.functional.gelu(acc)
        elif epilogue == EPI_RESIDUAL:
            if p_drop > 0:
                acc = acc * keep_scale(self._seed(seed), torch.arange(M)[:, None], torch.arange(N)[None, :], p_drop)
            acc = acc + v2(residual, M, N, ldr).float()
        elif epilogue == EPI_DGELU:
            acc = acc * gelu_grad(v2(aux, M, N, ldx).float())
        elif epilogue == EPI_TANH:
            acc = torch.tanh(acc)
        elif epilogue == EPI_GELU_DG:         # the derivative is saved instead of the pre-activation
            v2(aux, M, N, ldx).copy_(gelu_grad(acc))
            acc = torch.nn.functional.gelu(acc)
        elif epilogue == EPI_MULAUX:
            acc = acc * v2(aux, M, N, ldx).float()
        elif epilogue == EPI_ROWMAX:           # no C: per row and 64-column segment {max, sum exp(x - max), argmax bits, 0} -> aux
            assert N % 64 == 0
            seg = acc.view(M, N // 64, 64)
            mx, am = seg.max(-1)
            se = torch.exp(seg - mx[..., None]).sum(-1)
            idx = (am + torch.arange(N // 64)[None, :] * 64).to(torch.int32)
            rec = torch.stack([mx, se, idx.view(torch.float32), torch.zeros_like(mx)], -1)     # [M, nseg, 4]
            aux.view(-1)[:(N // 64) * M * 4].copy_(rec.permute(1, 0, 2).reshape(-1))
            return
        c = v2(C, M, N, ldc)
        if out_f32:
            assert C.dtype == torch.float32
        if accumulate:
            c.add_(acc)
        else:
            c.copy_(acc)
        if colsum is not None:
            torch.as_strided(colsum, (N,), (1,)).add_(c.float().sum(0))

    def set_deferred_reduce(self, on):
        pass                                   # the host restatement always reduces at once

    def flush_reductions(self):
        pass

    def flush_reductions_on(self, producer_stream):
        pass

    def wgrad_group_one_writer(self, problems):
        return True

    def take_f32(self, src, idx, own_lo, own_hi, dst):
        i = idx.long()
        own = (i >= own_lo) & (i < own_hi)
        dst[:i.numel()].copy_(torch.where(own, src[i], torch.zeros((), dtype=src.dtype)))

    def put_f32(self, dst, idx, src):
        dst[idx.long()] = src[:idx.numel()]

    def gemm_pair(self, c0, c1):
        """xl_gemm_pair: by contract the two xl_gemm calls"""
        self.gemm(*c0.a, **c0.kw)
        self.gemm(*c1.a, **c1.kw)

    def gemm_wgrad_group(self, problems, overwrite_mask=0):
        for i, (A, B, C, M, N, K, lda, ldb, ldc) in enumerate(problems):
            self.gemm(A, B, C, None, None, None, M, N, K, lda, ldb, ldc, a_kmajor=0, b_kmajor=0, out_f32=True,
                      accumulate=0 if (overwrite_mask >> i) & 1 else 1)

    @staticmethod
    def _ln(x, g, b, eps):
        mean = x.mean(1, keepdim=True)
        var = ((x - mean) ** 2).mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + eps)
        return (x - mean) * rstd * g + b, mean[:, 0], rstd[:, 0]

    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, M, N, eps):
        o, m, r = self._ln(v2(x, M, N, N).float(), gamma.float(), beta.float(), eps)
        v2(y, M, N, N).copy_(o)
        mean[:M].copy_(m)
        rstd[:M].copy_(r)

    @staticmethod
    def _ln_bwd(dy, x, g, mean, rstd):
        xh = (x - mean[:, None]) * rstd[:, None]
        gd = dy * g
        c1 = gd.mean(1, keepdim=True)
        c2 = (gd * xh).mean(1, keepdim=True)
        dx = rstd[:, None] * (gd - c1 - xh * c2)
        return dx, (dy * xh).sum(0), dy.sum(0)

    def workspace_floats(self, N):
        return 4096 * N

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias_prev, M, N, ws=None, dx_dropped=None,
                      p_drop=0.0, seed=0):
        d, dg, db = self._ln_bwd(v2(dy, M, N, N).float(), v2(x, M, N, N).float(), gamma.float(), mean[:M], rstd[:M])
        v2(dx, M, N, N).copy_(d)
        dgamma.add_(dg)
        dbeta.add_(db)
        if dx_dropped is not None and p_drop > 0:
            d = d * keep_scale(self._seed(seed), torch.arange(M)[:, None], torch.arange(N)[None, :], p_drop)                         # the kernel masks the fp32 value, then rounds
            v2(dx_dropped, M, N, N).copy_(d)
        if dbias_prev is not None:
            dbias_prev.add_(d.sum(0))

    def visn_ln_fwd(self, xv, pos, wbox, bbox, gv, bv, gb, bb, y, mean_v, rstd_v, mean_b, rstd_b, M, N, P, eps):
        a, mv, rv = self._ln(v2(xv, M, N, N).float(), gv, bv, eps)
        box = pos.view(M, P) @ wbox.view(N, P).t() + bbox
        b, mb, rb = self._ln(box, gb, bb, eps)
        v2(y, M, N, N).copy_((a + b) / 2)
        mean_v.copy_(mv); rstd_v.copy_(rv); mean_b.copy_(mb); rstd_b.copy_(rb)

    def visn_ln_bwd(self, dy, xv, pos, wbox, bbox, gv, gb, mean_v, rstd_v, mean_b, rstd_b, dxv, dgv, dbv, dgb, dbb,
                    dwbox, dbbox, dbias_visn, M, N, P, ws=None):
        dh = v2(dy, M, N, N).float() * 0.5
        d1, dg1, db1 = self._ln_bwd(dh, v2(xv, M, N, N).float(), gv, mean_v, rstd_v)
        box = pos.view(M, P) @ wbox.view(N, P).t() + bbox
        d2, dg2, db2 = self._ln_bwd(dh, box, gb, mean_b, rstd_b)
        v2(dxv, M, N, N).copy_(d1)
        dgv.add_(dg1); dbv.add_(db1); dgb.add_(dg2); dbb.add_(db2)
        dwbox.add_(d2.t() @ pos.view(M, P))
        dbbox.add_(d2.sum(0))
        if dbias_visn is not None:
            dbias_visn.add_(d1.sum(0))

    def embed_ln_fwd(self, ids, tt, word, pos, type_, gamma, beta, y, pre, mean, rstd, B, L, N, eps):
        M = B * L
        p = (word[ids.view(-1)].float() + pos[torch.arange(L).repeat(B)].float() + type_[tt.view(-1)].float())
        v2(pre, M, N, N).copy_(p)
        o, m, r = self._ln(v2(pre, M, N, N).float(), gamma, beta, eps)
        v2(y, M, N, N).copy_(o)
        mean.copy_(m); rstd.copy_(r)

    def embed_bwd(self, dpre, ids, tt, dword, dpos, dtype_tab, B, L, N, order=None, n_types=2):
        M = B * L
        d = v2(dpre, M, N, N).float()
        if order is not None:              # the loader's stable argsort of the ids: rows sorted by (id, row)
            key = ids.view(-1)[order.long()]
            assert sorted(order.tolist()) == list(range(M)) and bool((key[1:] >= key[:-1]).all())
            same = key[1:] == key[:-1]
            assert bool((order[1:][same] > order[:-1][same]).all())
        idf = ids.view(-1)
        ttf = tt.view(-1) if tt is not None else torch.zeros_like(idf)
        lf = torch.arange(L).repeat(B)
        dword.index_add_(0, idf[idf != 0], d[idf != 0])
        dpos.index_add_(0, lf[lf != 0], d[lf != 0])
        dtype_tab.index_add_(0, ttf[ttf != 0], d[ttf != 0])

    def codebook_gather(self, cluster_ids, vis_mask, centroids, mask_feat, feats, M, F):
        f = centroids[cluster_ids.view(-1)].float()
        if vis_mask is not None:
            f = torch.where(vis_mask.view(-1, 1) != 0, mask_feat.view(1, -1), f)
        v2(feats, M, F, F).copy_(f)

    def masked_colsum(self, x, mask, out, M, N, ldx, ws=None):
        xx = v2(x, M, N, ldx).float()
        out[:N].add_((xx * (mask.view(-1, 1) != 0)).sum(0))

    def colsum(self, x, out, M, N, ldx, ws=None):
        torch.as_strided(out, (N,), (1,)).add_(v2(x, M, N, ldx).float().sum(0))

    def dropout(self, x, y, M, N, ldx, ldy, p_drop, seed):
        v2(y, M, N, ldy).copy_(v2(x, M, N, ldx).float() * keep_scale(self._seed(seed), torch.arange(M)[:, None], torch.arange(N)[None, :], p_drop))

    def gelu_bwd(self, dy, pre, dx, n):
        dx.view(-1)[:n].copy_(dy.reshape(-1)[:n].float() * gelu_grad(pre.reshape(-1)[:n].float()))

    def tanh_bwd(self, dy, y, dx, n):
        yf = y.reshape(-1)[:n].float()
        dx.view(-1)[:n].copy_(dy.reshape(-1)[:n].float() * (1.0 - yf * yf))

    def bce_logits_fwd_bwd(self, logits, targets, dlogits, loss, M, N, ld_logits, ld_targets, ld_dlogits):
        x, t = v2(logits, M, N, ld_logits).float(), v2(targets, M, N, ld_targets).float()
        loss[0] += torch.nn.functional.binary_cross_entropy_with_logits(x, t)
        if dlogits is not None:
            d = v2(dlogits, M, ld_dlogits, ld_dlogits)
            d.zero_()
            d[:, :N].copy_((torch.sigmoid(x) - t) / float(M * N))

    def remask_lowest(self, prob, vis_mask, B, V, n_mask):
        p = prob.reshape(B, V)
        order = torch.argsort(p, dim=1, stable=True)           # ascending, ties -> lower index first
        m = torch.zeros(B, V, dtype=torch.uint8)
        if n_mask > 0:
            m.scatter_(1, order[:, :n_mask], 1)
        vis_mask.view(B, V).copy_(m)

    def sampler_update(self, pred_ids, vis_mask, code_ids, n):
        m = vis_mask.reshape(-1)[:n] != 0
        code_ids.view(-1)[:n][m] = pred_ids.reshape(-1)[:n][m].to(code_ids.dtype)

    def sampler_ar_update(self, prob, pred_ids, visited, vis_mask, code_ids, B, V, fixed_pos=-1):
        p, pr = prob.reshape(B, V), pred_ids.reshape(B, V)
        if fixed_pos >= 0:
            pos = torch.full((B,), fixed_pos, dtype=torch.long)
        else:
            q = p.masked_fill(visited.view(B, V) != 0, -10000.0)
            pos = (q == q.max(1, keepdim=True).values).float().argmax(1)          # first index of the maximum
            visited.view(B, V)[torch.arange(B), pos] = 1
        r = torch.arange(B)
        code_ids.view(B, V)[r, pos] = pr[r, pos].to(code_ids.dtype)
        vis_mask.view(B, V)[r, pos] = 0

    @staticmethod
    def _heads(t, B, n, H, dh, ld):
        return torch.as_strided(t, (B, H, n, dh), (n * ld, dh, ld, 1))

    @staticmethod
    def _pmask(B, H, nq, nk, p_drop, seed):
        if p_drop == 0:
            return 1.0
        row = torch.arange(B * H * nq).view(B, H, nq, 1)                # (b*H+h)*nq+q
        return keep_scale(seed, row, torch.arange(nk).view(1, 1, 1, nk), p_drop)

    # packed rows (include/xlxmert_hip.h xl_sdpa_*: q_rowoff / k_rowoff): unpack into the dense [B, H, n, dh] layout (zeros beyond
    # an example's length), compute as ever with the missing keys masked, store the real rows back, zero the pad tail
    def _load(self, t, B, n, H, dh, ld, off):
        if off is None:
            return self._heads(t, B, n, H, dh, ld).float(), None
        off = [int(x) for x in off.view(-1)[:B + 1]]
        mat = torch.as_strided(t, (off[B], H * dh), (ld, 1)).float()
        out, valid = torch.zeros(B, n, H * dh), torch.zeros(B, n, dtype=torch.bool)
        for b_ in range(B):
            m = min(n, off[b_ + 1] - off[b_])
            out[b_, :m] = mat[off[b_]:off[b_] + m]
            valid[b_, :m] = True
        return out.view(B, n, H, dh).permute(0, 2, 1, 3), valid

    def _store(self, dst, val, B, n, H, dh, ld, off, pad):
        if off is None:
            self._heads(dst, B, n, H, dh, ld).copy_(val)
            return
        off = [int(x) for x in off.view(-1)[:B + 1]]
        mat = torch.as_strided(dst, (max(pad, off[B]), H * dh), (ld, 1))
        rows = val.permute(0, 2, 1, 3).reshape(B, n, H * dh)
        for b_ in range(B):
            m = min(n, off[b_ + 1] - off[b_])
            mat[off[b_]:off[b_] + m].copy_(rows[b_, :m])
        if pad > off[B]:
            mat[off[B]:pad].zero_()

    def sdpa_keep_bits_bytes(self, B, H, nq, nk, dh):
        return 0                        # (the host restatement evaluates the mask hash wherever it needs the mask)

    def sdpa_fwd(self, q, k, v, key_mask, o, lse, B, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop=0.0, seed=0,
                 q_off=None, k_off=None, q_pad=0, k_pad=0, keep_bits=None):
        (Q, qv), (K_, kv), (V_, _) = (self._load(t, B, n, H, dh, ld, off)
                                      for t, n, ld, off in ((q, nq, ldq, q_off), (k, nk, ldk, k_off), (v, nk, ldv, k_off)))
        s = Q @ K_.transpose(-1, -2) * scale
        if key_mask is not None:
            s = s.masked_fill(key_mask.view(B, 1, 1, nk) == 0, float("-inf"))
        if kv is not None:
            s = s.masked_fill(~kv.view(B, 1, 1, nk), float("-inf"))
        lse.view(B, H, nq).copy_(torch.logsumexp(s, -1))
        out = (torch.softmax(s, -1) * self._pmask(B, H, nq, nk, p_drop, self._seed(seed))) @ V_
        self._store(o, out, B, nq, H, dh, ldo, q_off, q_pad)

    def attn_probs(self, q, k, key_mask, lse, probs, B, H, nq, nk, dh, ldq, ldk, scale, p_drop=0.0, seed=0, q_off=None, k_off=None):
        (Q, qv), (K_, kv) = self._load(q, B, nq, H, dh, ldq, q_off), self._load(k, B, nk, H, dh, ldk, k_off)
        p = torch.exp(Q @ K_.transpose(-1, -2) * scale - lse.view(B, H, nq, 1))
        if key_mask is not None:
            p = p.masked_fill(key_mask.view(B, 1, 1, nk) == 0, 0.0)
        if kv is not None:
            p = p.masked_fill(~kv.view(B, 1, 1, nk), 0.0)
        if qv is not None:
            p = p.masked_fill(~qv.view(B, 1, nq, 1), 0.0)
        p = torch.nan_to_num(p, nan=0.0, posinf=0.0)
        probs.view(B, H, nq, nk).copy_(p * self._pmask(B, H, nq, nk, p_drop, self._seed(seed)))

    def sdpa_bwd(self, q, k, v, key_mask, dout, lse, dq, dk, dv, B, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq, lddk,
                 lddv, scale, p_drop=0.0, seed=0, bias_grad=None, ws=None, q_off=None, k_off=None, q_pad=0, k_pad=0, keep_bits=None):
        (Q, qv), (K_, kv), (V_, _), (dO, _) = (self._load(t, B, n, H, dh, ld, off) for t, n, ld, off in
                                               ((q, nq, ldq, q_off), (k, nk, ldk, k_off), (v, nk, ldv, k_off), (dout, nq, ldo, q_off)))
        s = Q @ K_.transpose(-1, -2) * scale
        p = torch.exp(s - lse.view(B, H, nq, 1))
        if key_mask is not None:
            p = p.masked_fill(key_mask.view(B, 1, 1, nk) == 0, 0.0)
        if kv is not None:
            p = p.masked_fill(~kv.view(B, 1, 1, nk), 0.0)
        if qv is not None:                                   # queries beyond an example's length do not exist
            p = p.masked_fill(~qv.view(B, 1, nq, 1), 0.0)
        p = torch.nan_to_num(p, nan=0.0, posinf=0.0)
        msk = self._pmask(B, H, nq, nk, p_drop, self._seed(seed))
        dp = (dO @ V_.transpose(-1, -2)) * msk
        delta = (p * dp).sum(-1, keepdim=True)
        ds = p * (dp - delta) * scale
        p = p * msk
        gq, gk, gv = ds @ K_, ds.transpose(-1, -2) @ Q, p.transpose(-1, -2) @ dO
        self._store(dq, gq, B, nq, H, dh, lddq, q_off, q_pad)
        self._store(dk, gk, B, nk, H, dh, lddk, k_off, k_pad)
        self._store(dv, gv, B, nk, H, dh, lddv, k_off, k_pad)
        if bias_grad is not None:
            HD = H * dh
            for i, g in enumerate((gq, gk, gv)):
                bias_grad[i * HD:(i + 1) * HD] += g.sum(dim=(0, 2)).reshape(HD)       # [B,H,n,dh] -> [H*dh]

    def mask_counts(self, labels, vis_mask, counts, nmask, B, V):
        counts[0] = (labels != -100).sum().float()
        nmask.copy_((vis_mask.view(B, V) != 0).sum(1).float())

    def ce_fwd_bwd(self, logits, labels, counts, dlogits, loss_out, row_lse, row_argmax, row_maxprob, M, K, ldl, lddl,
                   grad_scale=1.0):
        lg = v2(logits, M, K, ldl)
        lse = torch.logsumexp(lg, 1)
        if row_lse is not None:
            row_lse.copy_(lse)
        if row_argmax is not None:
            row_argmax.copy_(lg.argmax(1).int())
        if row_maxprob is not None:
            row_maxprob.copy_(torch.exp(lg.max(1).values - lse))
        if labels is None:
            return
        lab = labels.view(-1)
        valid = lab != -100
        cnt = counts[0].clamp(min=1)
        safe = lab.clamp(min=0)
        nll = (lse - lg.gather(1, safe[:, None])[:, 0]) * valid
        if loss_out is not None:
            loss_out[0] += nll.sum() / cnt
        if dlogits is not None:
            g = torch.softmax(lg, 1)
            g[torch.arange(M), safe] -= 1.0
            g = g * valid[:, None] * (grad_scale / cnt)
            v2(dlogits, M, K, lddl).copy_(g)

    def featloss_fwd_bwd(self, pred, centroids, cluster_ids, vis_mask, nmask, dpred, loss_out, B, V, F, grad_scale=1.0,
                         rows=None, n_rows=0, targets=None):
        g = torch.arange(B * V) if rows is None else rows.view(-1)[:n_rows].long()
        pad = g < 0                      # padding entries of the row list: no loss, zero gradient row
        g = g.clamp(min=0)
        M = g.numel()
        p = v2(pred, M, F, F).float()
        t = (centroids[cluster_ids.view(-1)[g]] if targets is None else targets.view(B * V, F)[g]).float()
        d = p - t
        sl1 = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5).mean(1)
        w = ((vis_mask.view(-1) != 0).float() / (nmask.clamp(min=1).repeat_interleave(V) * B))[g]
        w = torch.where(pad, torch.zeros_like(w), w)
        if loss_out is not None:
            loss_out[0] += (w * sl1).sum()
        if dpred is not None:
            v2(dpred, M, F, F).copy_(grad_scale * w[:, None] / F * d.clamp(-1, 1))

    def gather_rows(self, src, rows, dst, n_rows, N, ld_src, ld_dst):
        g = rows.view(-1)[:n_rows].long()
        out = torch.as_strided(src, (int(g.max()) + 1, N), (ld_src, 1))[g.clamp(min=0)].clone()
        out[g < 0] = 0                   # padding entries: zero rows
        v2(dst, n_rows, N, ld_dst).copy_(out)

    def scatter_rows(self, src, rows, dst, n_rows, N, ld_src, ld_dst):
        g = rows.view(-1)[:n_rows].long()
        keep = g >= 0                    # padding entries are skipped
        torch.as_strided(dst, (int(g.max()) + 1, N), (ld_dst, 1))[g[keep]] = v2(src, n_rows, N, ld_src)[keep]

    def rowmax_combine(self, ws, n_seg, M, row_maxprob, row_argmax, row_lse=None):
        rec = ws.view(-1)[:n_seg * M * 4].view(n_seg, M, 4)
        mx, se, idx = rec[..., 0], rec[..., 1], rec[..., 2].contiguous().view(torch.int32)
        gmx = mx.max(0).values
        tot = (se * torch.exp(mx - gmx[None, :])).sum(0)
        cand = torch.where(mx == gmx[None, :], idx, torch.full_like(idx, 2 ** 31 - 1))
        if row_argmax is not None:
            row_argmax[:M].copy_(cand.min(0).values)
        if row_maxprob is not None:
            row_maxprob[:M].copy_(1.0 / tot)
        if row_lse is not None:
            row_lse[:M].copy_(gmx + torch.log(tot))

    def gather_labels(self, labels, rows, out, n_rows):
        g = rows.view(-1)[:n_rows].long()
        out.view(-1)[:n_rows].copy_(torch.where(g >= 0, labels.view(-1)[g.clamp(min=0)], torch.full_like(g, -100)))

    def sumsq(self, g, out, n, scratch=None):
        out[0] += (g[:n].double() ** 2).sum().float()

    def schedule_step(self, step, base_lr, warmup_steps, total_steps, beta1, beta2, lr_and_steps):
        step[0] += 1
        t = int(step[0])
        done = t - 1
        f = done / max(1, warmup_steps) if done < warmup_steps else max(0.0, (total_steps - done) / max(1, total_steps - warmup_steps))
        lr_and_steps[0], lr_and_steps[1], lr_and_steps[2], lr_and_steps[3] = base_lr * f, 1.0 - beta1 ** t, 1.0 - beta2 ** t, float(t)

    def adamw(self, p, g, m, v, p_compute, decay_flags, sumsq, lr_and_steps, n, beta1, beta2, eps, weight_decay,
              max_norm, grad_scale=1.0, chunk_steps=None, zero_grad=False):
        lr, bc1, bc2 = (float(x) for x in lr_and_steps[:3])
        clip = grad_scale
        if max_norm > 0 and sumsq is not None:
            norm = math.sqrt(float(sumsq[0])) * grad_scale
            clip *= min(1.0, max_norm / (norm + 1e-6))
        fl = decay_flags.repeat_interleave(256)[:n] if decay_flags is not None else torch.zeros(n, dtype=torch.uint8)
        act = (fl & 2) == 0                                    # bit 1: tensor without a gradient this step -> untouched
        gg = g[:n] * clip
        m_new = m[:n] * beta1 + gg * (1 - beta1)
        v_new = v[:n] * beta2 + gg * gg * (1 - beta2)
        if chunk_steps is not None:
            t = chunk_steps.repeat_interleave(256)[:n].double().clamp(min=1)
            step = (lr * torch.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)).float()
        else:
            step = torch.full((n,), lr * math.sqrt(bc2) / bc1)
        p_new = p[:n] - step * (m_new / (v_new.sqrt() + eps))
        if weight_decay > 0:
            dec = (fl & 1) != 0
            p_new = torch.where(dec, p_new - lr * weight_decay * p_new, p_new)
        m[:n].copy_(torch.where(act, m_new, m[:n]))
        v[:n].copy_(torch.where(act, v_new, v[:n]))
        p[:n].copy_(torch.where(act, p_new, p[:n]))
        if p_compute is not None and p_compute.data_ptr() != p.data_ptr():
            p_compute[:n].copy_(p[:n])
        if zero_grad:
            # bit 2: "the next backward overwrites this chunk": the kernel leaves such a gradient as it is; the host restatement
            # POISONS it, so that any path that reads or accumulates into a kept chunk without overwriting it first shows up as NaN
            keep = (fl & 4) != 0
            cleared = torch.where(keep, torch.full_like(g[:n], float("nan")), torch.zeros_like(g[:n]))
            g[:n].copy_(torch.where(act, cleared, g[:n]))

    def cast_from_f32(self, src, dst, n):
        if dst.data_ptr() != src.data_ptr():
            dst.view(-1)[:n].copy_(src.view(-1)[:n])

    def cast_to_f32(self, src, dst, n):
        dst.view(-1)[:n].copy_(src.view(-1)[:n])
