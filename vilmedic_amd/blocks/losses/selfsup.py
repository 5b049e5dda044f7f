"""ConVIRT / InfoNCE / GLoRIA losses on the HIP path (same constructors and return tuples as the reference).

ref: vilmedic/blocks/losses/selfsup/ConVIRTLoss.py:5-38, InfoNCELoss.py:5-24, GLoRIALoss.py:5-170.

The [B,B] similarity S = n(a) n(b)^T / tau is computed tile by tile on the MFMA: row / column log-sum-exp and the diagonal in the
forward pass, which also parks the fp32 tiles in the workspace; the backward pass streams them once into the gradient matrix G
(csrc/contrastive.hip), forms dA = G B / tau and dB = G^T A / tau as GEMM tiles of one launch and applies the normalisation backward as a
row pass.  When torch.distributed is initialised with world_size > 1 the text / image embeddings are
all-gathered first (RCCL), so every rank sees the GLOBAL batch of negatives (SURVEY §8e -- a capability the reference
lacks: under DDP it contrasts within the local shard only, conVIRT.py:97-100).
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import check, lib, ptr, stream

BF16 = torch.bfloat16


def _pad8(n):
    return (n + 7) // 8 * 8


class _SimilarityLossFn(torch.autograd.Function):
    """(a [R,D], b [C,D]) -> per-row losses  row_i = lse_j S_ij - S_{i,i+off},  col_i = lse_j S_{j,i+off} - S_{i,i+off}  with
    S = n(a) n(b)^T * inv_tau  (n = L2 normalisation when ``normalize``); ``diag_offset`` = off pairs row i with column i + off
    (a rank's local rows against the gathered columns).  Two C-ABI calls, six short kernel launches forward + backward
    (csrc/contrastive.hip); the workspace carries the fp32 similarity from the forward call to the backward call."""

    @staticmethod
    def forward(ctx, a, b, normalize, inv_tau, eps, diag_offset=0):
        R, D = a.shape
        Cn = b.shape[0]
        if D % 8:
            raise ValueError("embedding dim must be a multiple of 8")
        dev = a.device
        a32, b32 = a.detach().float().contiguous(), b.detach().float().contiguous()
        L = lib()
        f32 = dict(dtype=torch.float32, device=dev)
        ah = torch.empty(R, D, dtype=BF16, device=dev)
        bh = torch.empty(Cn, D, dtype=BF16, device=dev)
        stats = torch.empty(3 * (R + Cn), **f32)         # norms | log-sum-exps | losses, rows then columns in each third
        na, nb, lse_r, lse_c, loss_r, loss_c = stats.split([R, Cn, R, Cn, R, Cn])
        ws = torch.empty(L.vm_contrastive_ws(R, Cn), dtype=torch.uint8, device=dev)
        check(L.vm_contrastive_loss_fwd(ptr(a32), ptr(b32), R, Cn, D, int(normalize), eps, inv_tau, int(diag_offset), ptr(ah), ptr(bh), ptr(na), ptr(nb),
                                        ptr(lse_r), ptr(lse_c), ptr(loss_r), ptr(loss_c), ptr(ws), ws.numel(), stream()), "vm_contrastive_loss_fwd")
        ctx.save_for_backward(a32, b32, ah, bh, stats, ws)
        ctx.meta = (normalize, inv_tau, eps, R, Cn, D, int(diag_offset))
        # the pairs (i, i + off) that exist: rows lo..hi-1 and the columns they are paired with
        lo, hi = max(0, -diag_offset), min(R, Cn - diag_offset)
        ctx.span = (lo, hi)
        return loss_r[lo:hi], loss_c[lo + diag_offset:hi + diag_offset]

    @staticmethod
    def backward(ctx, g_row, g_col):
        a32, b32, ah, bh, stats, ws = ctx.saved_tensors
        normalize, inv_tau, eps, R, Cn, D, off = ctx.meta
        lo, hi = ctx.span
        na, nb, lse_r, lse_c, _, _ = stats.split([R, Cn, R, Cn, R, Cn])
        dev = ah.device
        g = torch.zeros(R + Cn, dtype=torch.float32, device=dev)
        if g_row is not None:
            g[lo:hi] = g_row.float()
        if g_col is not None:
            g[R + lo + off:R + hi + off] = g_col.float()
        d = torch.empty(R + Cn, D, dtype=torch.float32, device=dev)
        check(lib().vm_contrastive_loss_bwd(ptr(a32), ptr(b32), ptr(ah), ptr(bh), ptr(na), ptr(nb), R, Cn, D, int(normalize), eps, inv_tau, off,
                                            ptr(lse_r), ptr(lse_c), ptr(g[:R]), ptr(g[R:]), ptr(d[:R]), ptr(d[R:]), ptr(ws), ws.numel(), stream()),
              "vm_contrastive_loss_bwd")
        return d[:R], d[R:], None, None, None, None


def _maybe_gather(*xs):
    """global negatives under data parallelism -- in TRAINING only: validation shards may differ by one batch between ranks, and a
    collective inside the loss would then wait forever; under no_grad the loss contrasts within the local shard (what the reference
    does under DDP)"""
    from ...parallel import active, all_gather_with_grad
    dist = active()
    if dist is not None and torch.is_grad_enabled():
        n = xs[0].shape[0]
        r = dist.get_rank()
        return [all_gather_with_grad(x.contiguous(), dist) for x in xs], slice(r * n, (r + 1) * n), dist.get_world_size()
    return list(xs), slice(None), 1


def _paired_losses(first, second, normalize, inv_tau, eps=1e-8):
    """per-sample contrastive losses of the LOCAL pairs: (first -> second, second -> first).  Single process / evaluation: one problem on the
    [B, B] similarity.  Data parallel training: both embedding matrices are all-gathered (RCCL; backward = sum over ranks +
    own slice) and every rank evaluates only ITS rows of the global similarity -- two [b, B_global] problems whose paired column is
    rank * b + i -- instead of recomputing the whole [B_global, B_global] matrix on every rank (2 / world of the work).  The mean over the
    local rows, averaged over ranks by the gradient all-reduce, is the global-batch loss and its exact gradient."""
    from ...parallel import active, all_gather_with_grad
    dist = active()
    if dist is None or not torch.is_grad_enabled():
        return _SimilarityLossFn.apply(first, second, normalize, inv_tau, eps)
    off = dist.get_rank() * first.shape[0]
    fg, sg = all_gather_with_grad(first.contiguous(), dist), all_gather_with_grad(second.contiguous(), dist)
    lf, _ = _SimilarityLossFn.apply(first, sg, normalize, inv_tau, eps, off)       # rows of S:   first_i  against every second_j
    ls, _ = _SimilarityLossFn.apply(second, fg, normalize, inv_tau, eps, off)      # rows of S^T: second_i against every first_j
    return lf, ls


class ConVIRTLoss(nn.Module):
    def __init__(self, tau, lambda_, **kwargs):
        super().__init__()
        self.tau = tau
        self.lambda_ = lambda_

    def forward(self, linguistic, visual):
        loss_l, loss_v = _paired_losses(linguistic, visual, True, 1.0 / self.tau)
        loss = torch.mean(self.lambda_ * loss_v + (1 - self.lambda_) * loss_l)
        return loss, loss_l, loss_v

    def __repr__(self):
        return "ConVIRTLoss(\n\t(cos_loss): CosineSimilarity()\n\t(tau): {}\n\t(lambda_): {}\n)".format(self.tau, self.lambda_)


class InfoNCELoss(nn.Module):
    """raw dot-product logits, CE both ways; ``tau`` is stored but NOT applied -- exactly as the reference."""

    def __init__(self, tau, **kwargs):
        super().__init__()
        self.tau = tau

    def forward(self, linguistic, visual):
        loss_t, loss_i = _paired_losses(linguistic, visual, False, 1.0)
        loss = ((loss_i + loss_t) / 2).mean()
        return loss, loss_t, loss_i

    def __repr__(self):
        return "InfoNCELoss(\n\t(tau): {}\n)".format(self.tau)


# ----------------------------------------------------------------------------- VICReg
class _CovOffDiagFn(torch.autograd.Function):
    """z [N,D] -> sum of the squared OFF-diagonal entries of cov(z) = zc^T zc / (N-1), zc = z - mean_0(z).  The [D,D] product
    (contraction over the batch rows) and its gradient dz = 4/(N-1) zc C_off are bf16 MFMA GEMMs with fp32 output."""

    @staticmethod
    def forward(ctx, z):
        N, D = z.shape
        if D % 8:
            raise ValueError("embedding dim must be a multiple of 8")
        zc = z.detach().float()
        zc = zc - zc.mean(0)
        Np = _pad8(N)
        zh = torch.zeros(Np, D, dtype=BF16, device=z.device)
        zh[:N] = zc
        cov = torch.empty(D, D, dtype=torch.float32, device=z.device)
        ops.gemm(zh, 1, zh, 1, cov, D, D, Np, alpha=1.0 / (N - 1))
        cov.diagonal().zero_()
        ctx.save_for_backward(zh, cov)
        ctx.N = N
        return cov.pow(2).sum()

    @staticmethod
    def backward(ctx, g):
        zh, cov = ctx.saved_tensors
        N, (Np, D) = ctx.N, zh.shape
        dz = torch.empty(Np, D, dtype=torch.float32, device=zh.device)
        ops.gemm(zh, 0, cov.to(BF16), 0, dz, Np, D, D, alpha=4.0 / (N - 1))
        dz = dz[:N]
        return (dz - dz.mean(0)) * g


class VICREGLoss(nn.Module):
    """ref: vilmedic/blocks/losses/selfsup/VICREGLoss.py:6-99 -- invariance (MSE) + variance hinge on the per-dimension std +
    covariance (squared off-diagonal entries of each view's [D,D] covariance, / D)."""

    def __init__(self, sim_loss_weight=25.0, var_loss_weight=25.0, cov_loss_weight=1.0, **kwargs):
        super().__init__()
        self.sim_loss_weight, self.var_loss_weight, self.cov_loss_weight = sim_loss_weight, var_loss_weight, cov_loss_weight

    def forward(self, z1, z2):
        return (self.sim_loss_weight * self.invariance_loss(z1, z2) + self.var_loss_weight * self.variance_loss(z1, z2)
                + self.cov_loss_weight * self.covariance_loss(z1, z2))

    @staticmethod
    def invariance_loss(z1, z2):
        return nn.functional.mse_loss(z1.float(), z2.float())

    @staticmethod
    def variance_loss(z1, z2):
        eps = 1e-4
        std1, std2 = torch.sqrt(z1.float().var(dim=0) + eps), torch.sqrt(z2.float().var(dim=0) + eps)
        return torch.mean(torch.relu(1 - std1)) + torch.mean(torch.relu(1 - std2))

    @staticmethod
    def covariance_loss(z1, z2):
        D = z1.shape[1]
        return _CovOffDiagFn.apply(z1) / D + _CovOffDiagFn.apply(z2) / D

    def __repr__(self):
        return ("VICREGLoss(\n\t(sim_loss_weight): {}\n\t(var_loss_weight): {}\n\t(cov_loss_weight): {}\n)"
                .format(self.sim_loss_weight, self.var_loss_weight, self.cov_loss_weight))


# ----------------------------------------------------------------------------- GLoRIA (global: HIP similarity; local: batched torch ops)
def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return (w12 / (w1 * w2).clamp(min=eps)).squeeze()


def gloria_attention_fn(query, context, temp1):
    """ref: GLoRIALoss.py:13-51 (query [B,D,T], context [B,D,ih,iw])."""
    B, T = query.size(0), query.size(2)
    ih, iw = context.size(2), context.size(3)
    S = ih * iw
    ctx = context.view(B, -1, S)
    attn = torch.bmm(ctx.transpose(1, 2), query)
    attn = torch.softmax(attn.view(B * S, T), dim=-1).view(B, S, T)
    attn = attn.transpose(1, 2).contiguous().view(B * T, S) * temp1
    attn = torch.softmax(attn, dim=-1).view(B, T, S)
    return torch.bmm(ctx, attn.transpose(1, 2)), attn.view(B, -1, ih, iw)


def _pad16(n):
    return (n + 15) // 16 * 16


def _pad_k(n):
    """contraction lengths: a multiple of 64 keeps the GEMM on the LDS-DMA kernels (K % 64 == 0, also after the x 3 of the split
    operands); tiny test shapes stay at multiples of 16 (register-staged kernel)"""
    return (n + 63) // 64 * 64 if n >= 64 else _pad16(n)


def _transpose(src, sbs, lds, dst, dbs, ldd, batch, rows, cols, drows, dcols):
    check(lib().vm_transpose_f32(ptr(src), sbs, lds, ptr(dst), dbs, ldd, batch, rows, cols, drows, dcols, stream()), "vm_transpose_f32")


def _split3(src2d, role, along_rows, block):
    """fp32 [rows, cols] -> the (hi, hi, lo) / (hi, lo, hi) bf16 parts of a "bf16 x 3" GEMM operand (vm_split3_bf16), laid out along the
    contraction axis: [rows, 3 cols] (along_rows = 0) or [3 rows, cols]"""
    rows, cols = src2d.shape
    dst = torch.empty((3 * rows, cols) if along_rows else (rows, 3 * cols), dtype=BF16, device=src2d.device)
    check(lib().vm_split3_bf16(ptr(src2d), src2d.stride(0), rows, cols, ptr(dst), dst.stride(0), role, int(along_rows), block, stream()),
          "vm_split3_bf16")
    return dst


_A, _B = 0, 1       # operand roles of vm_split3_bf16


class _GloriaLocalFn(torch.autograd.Function):
    """(local image features [B,D,ih,iw], word embeddings [B,D,T], caption lengths) -> (loss0, loss1, a2) with
    loss0 = CE(sims, arange), loss1 = CE(sims^T, arange), sims [B img, B cap] the word-region matching score of every pair and
    a2 [B img, B*Tp, Pp] the attention of every word over every image's regions (csrc/gloria.hip; layouts in include/vmhip.h).
    The six contractions (S, the context vectors and their four gradient products) run on the bf16 MFMA GEMM with "bf16 x 3" split
    operands and fp32 output; everything between them is fp32."""

    @staticmethod
    def forward(ctx, img, words, lens, temp1, temp2, temp3):
        dev = img.device
        B, D, ih, iw = img.shape
        P, T = ih * iw, words.shape[2]
        Tmax = int(lens.max())
        if Tmax > T or int(lens.min()) < 1:
            raise ValueError("GLoRIA local loss: caption lengths must be in 1..T")
        Tp, Pp, Dp = _pad16(Tmax), _pad_k(P), _pad_k(D)
        while B * Tp >= 64 and (B * Tp) % 64:                   # B * Tp is a contraction length too (the gradients w.r.t. the features)
            Tp += 16
        f32 = dict(dtype=torch.float32, device=dev)
        L = lib()
        ctx3 = img.detach().float().reshape(B, D, P).contiguous()
        w3 = words.detach().float().contiguous()
        lens_dev = lens.to(dev)
        Ct = torch.empty(B * Pp, Dp, **f32)                     # (image, region) x features, zero padded to multiples of 16
        _transpose(ctx3, D * P, P, Ct, Pp * Dp, Dp, B, D, P, Pp, Dp)
        Wt = torch.empty(B * Tp, Dp, **f32)                     # (caption, word) x features (rows >= cap_lens[i] are masked in the kernels)
        _transpose(w3, D * T, T, Wt, Tp * Dp, Dp, B, D, Tmax, Tp, Dp)
        nw = torch.empty(B * Tp, **f32)
        check(L.vm_row_norm_f32(ptr(Wt), Dp, ptr(nw), B * Tp, Dp, stream()), "vm_row_norm_f32")
        # S[(i,t), (j,p)] = <word, region>
        Ct_B = _split3(Ct, _B, False, Dp)                       # [B*Pp, 3 Dp]
        S = torch.empty(B * Tp, B * Pp, **f32)
        ops.gemm(_split3(Wt, _A, False, Dp), 0, Ct_B, 0, S, B * Tp, B * Pp, 3 * Dp)
        a2 = torch.empty(B, B * Tp, Pp, **f32)
        dot = torch.empty(B * Tp, B, **f32)
        colstat = torch.empty(B, B, 2, Pp, **f32)
        check(L.vm_gloria_attn_fwd(ptr(S), S.stride(0), ptr(lens_dev), B, Tp, P, Pp, temp1, ptr(a2), ptr(dot), ptr(colstat), stream()),
              "vm_gloria_attn_fwd")
        # x_j = a2_j C_j  (contraction over the regions of image j)
        a2_A = _split3(a2.view(B * B * Tp, Pp), _A, False, Pp).view(B, B * Tp, 3 * Pp)
        Ct_Bs = _split3(Ct, _B, True, Pp).view(B, 3 * Pp, Dp)  # per image: [3 Pp, Dp]
        x = torch.empty(B, B * Tp, Dp, **f32)
        ops.gemm_grouped([(a2_A[j], Ct_Bs[j], x[j], B * Tp, Dp, 3 * Pp) for j in range(B)], 0, 1)
        del a2_A
        sims, simsT = torch.empty(B, B, **f32), torch.empty(B, B, **f32)
        cosv, nxv = torch.empty(B * Tp, B, **f32), torch.empty(B * Tp, B, **f32)
        check(L.vm_gloria_cos_fwd(ptr(x), Dp, ptr(nw), ptr(dot), ptr(lens_dev), B, Tp, Dp, temp2, temp3, 1e-8, ptr(sims), ptr(simsT), ptr(cosv),
                                  ptr(nxv), stream()), "vm_gloria_cos_fwd")
        labels = torch.arange(B, device=dev)
        losses = torch.zeros(2, **f32)
        dsims, dsimsT = torch.empty_like(sims), torch.empty_like(simsT)
        check(L.vm_ce_smooth_fwd_bwd(ptr(sims), ptr(labels), B, B, 0.0, ptr(losses[0:1]), ptr(dsims), 1.0 / B, stream()), "vm_ce_smooth_fwd_bwd")
        check(L.vm_ce_smooth_fwd_bwd(ptr(simsT), ptr(labels), B, B, 0.0, ptr(losses[1:2]), ptr(dsimsT), 1.0 / B, stream()), "vm_ce_smooth_fwd_bwd")
        losses = losses / B
        ctx.save_for_backward(S, a2, x, Wt, Ct, Ct_B, Ct_Bs, colstat, cosv, nxv, nw, sims, dsims, dsimsT, lens_dev)
        ctx.meta = (B, D, P, T, Tmax, Tp, Pp, Dp, ih, iw, temp1, temp2, temp3)
        ctx.mark_non_differentiable(a2)
        return losses[0], losses[1], a2

    @staticmethod
    def backward(ctx, g0, g1, _ga2):
        S, a2, x, Wt, Ct, Ct_B, Ct_Bs, colstat, cosv, nxv, nw, sims, dsims, dsimsT, lens_dev = ctx.saved_tensors
        B, D, P, T, Tmax, Tp, Pp, Dp, ih, iw, temp1, temp2, temp3 = ctx.meta
        dev = S.device
        f32 = dict(dtype=torch.float32, device=dev)
        L = lib()
        g0 = g0 if g0 is not None else torch.zeros((), **f32)
        g1 = g1 if g1 is not None else torch.zeros((), **f32)
        ds, dsT = (dsims * g0).contiguous(), (dsimsT * g1).contiguous()
        dx = torch.empty(B, B * Tp, Dp, **f32)
        dW = torch.empty(B * Tp, Dp, **f32)
        check(L.vm_gloria_cos_bwd(ptr(ds), ptr(dsT), ptr(sims), ptr(cosv), ptr(nxv), ptr(nw), ptr(x), ptr(Wt), Dp, ptr(lens_dev), B, Tp, Dp,
                                  temp2, temp3, 1e-8, ptr(dx), ptr(dW), stream()), "vm_gloria_cos_bwd")
        # through x_j = a2_j C_j:  d a2_j = d x_j C_j^T  (contraction over the features),  d C_j = a2_j^T d x_j  (over the B*Tp word rows)
        dx2 = dx.view(B * B * Tp, Dp)
        dx_A = _split3(dx2, _A, False, Dp).view(B, B * Tp, 3 * Dp)
        da2 = torch.empty(B, B * Tp, Pp, **f32)
        ops.gemm_grouped([(dx_A[j], Ct_B[j * Pp:(j + 1) * Pp], da2[j], B * Tp, Pp, 3 * Dp) for j in range(B)], 0, 0)
        del dx_A
        a2_As = _split3(a2.view(B * B * Tp, Pp), _A, True, B * Tp).view(B, 3 * B * Tp, Pp)
        dx_Bs = _split3(dx2, _B, True, B * Tp).view(B, 3 * B * Tp, Dp)
        dCt = torch.empty(B, Pp, Dp, **f32)
        ops.gemm_grouped([(a2_As[j], dx_Bs[j], dCt[j], Pp, Dp, 3 * B * Tp) for j in range(B)], 1, 1)
        del a2_As, dx_Bs, dx2, dx
        dS = torch.empty_like(S)
        check(L.vm_gloria_attn_bwd(ptr(S), S.stride(0), ptr(colstat), ptr(da2), ptr(lens_dev), B, Tp, P, Pp, temp1, ptr(dS), stream()),
              "vm_gloria_attn_bwd")
        # through S = Wt Ct^T:  d Wt += d S Ct  (over all regions),  d Ct += d S^T Wt  (over all word rows)
        ops.gemm(_split3(dS, _A, False, B * Pp), 0, _split3(Ct, _B, True, B * Pp), 1, dW, B * Tp, Dp, 3 * B * Pp, accumulate=True)
        ops.gemm(_split3(dS, _A, True, B * Tp), 1, _split3(Wt, _B, True, B * Tp), 1, dCt.view(B * Pp, Dp), B * Pp, Dp, 3 * B * Tp, accumulate=True)
        # back to the callers' layouts
        dCn = torch.empty(B, Dp, Pp, **f32)
        _transpose(dCt, Pp * Dp, Dp, dCn, Dp * Pp, Pp, B, Pp, Dp, Dp, Pp)
        d_img = dCn[:, :D, :P].reshape(B, D, ih, iw)
        dWn = torch.empty(B, Dp, Tp, **f32)
        _transpose(dW, Tp * Dp, Dp, dWn, Dp * Tp, Tp, B, Tp, Dp, Dp, Tp)
        d_words = torch.zeros(B, D, T, **f32)
        d_words[:, :, :Tmax] = dWn[:, :D, :Tmax]
        return d_img, d_words, None, None, None, None


class GLoRIALoss(nn.Module):
    # fp32 working set of the local loss is ~6 matrices of [B*Tp, B*Pp]: beyond this budget the local part falls back to the rank's shard
    LOCAL_GATHER_BYTES = 48 << 30

    def __init__(self, local_loss_weight=1.0, global_loss_weight=1.0, temp1=4.0, temp2=5.0, temp3=10.0, gather_local=True):
        """``gather_local`` (data parallel training only): True = every caption is contrasted with the region features of the GLOBAL
        batch while that fits ``LOCAL_GATHER_BYTES``; False = the local loss contrasts within the rank's shard --
        what the reference computes under DDP (ref:vilmedic/blocks/losses/selfsup/GLoRIALoss.py:78-129 sees the local batch only) --
        and only the two global embeddings are gathered."""
        super().__init__()
        self.local_loss_weight, self.global_loss_weight = local_loss_weight, global_loss_weight
        self.temp1, self.temp2, self.temp3 = temp1, temp2, temp3
        self.gather_local = gather_local

    def _gather_local_fits(self, local_features, word_embeddings):
        from ...parallel import active
        dist = active()
        if dist is None or not torch.is_grad_enabled():
            return True
        Bg = local_features.shape[0] * dist.get_world_size()
        Pp = _pad8(local_features.shape[2] * local_features.shape[3])
        Tp = _pad8(word_embeddings.shape[2])
        return 6 * 4 * (Bg * Tp) * (Bg * Pp) <= self.LOCAL_GATHER_BYTES

    def forward(self, global_features, local_features, word_embeddings, sent_embeddings, sents):
        # data parallel training: every caption is contrasted with the images of the GLOBAL batch (and vice versa), so the local
        # feature maps [b, D, 19, 19], the word embeddings [b, D, T] and both global embeddings are all-gathered (RCCL; backward =
        # sum over ranks + own slice) together with the word lists -- SURVEY §8e "GLoRIA local loss"
        if self.gather_local and self._gather_local_fits(local_features, word_embeddings):
            (global_features, local_features, word_embeddings, sent_embeddings), _, world = _maybe_gather(
                global_features, local_features, word_embeddings, sent_embeddings)
            from ...parallel import active
            if active() is not None and torch.is_grad_enabled():
                import torch.distributed as dist
                parts = [None] * world
                dist.all_gather_object(parts, list(sents))
                sents = [s_ for p_ in parts for s_ in p_]
        else:                    # local loss on the rank's shard (the reference's DDP behaviour); global loss over the gathered embeddings
            (global_features, sent_embeddings), _, world = _maybe_gather(global_features, sent_embeddings)
        cap_lens = [len([w for w in sent if not w.startswith("[")]) + 1 for sent in sents]
        l0, l1, attn_maps = self._local(local_features.float(), word_embeddings.float(), cap_lens)
        # global: cosine-sim [B,B] * temp3 -> CE both ways == the HIP similarity loss with inv_tau = temp3 (mean over rows)
        row, col = _SimilarityLossFn.apply(global_features.float(), sent_embeddings.float(), True, self.temp3, 1e-8)
        loss = (l0 + l1) * self.local_loss_weight + (row.mean() + col.mean()) * self.global_loss_weight
        return loss, attn_maps

    def _local(self, img, words, cap_lens):
        """ref: GLoRIALoss.py:78-129.  The reference loops over captions, repeating each caption B times and attending the whole image
        batch (two bmm's and two softmaxes per caption).  Here every (caption i, image j) pair is handled at once by the kernels of
        csrc/gloria.hip (forward and backward, fp32): -> (loss0, loss1, attention maps of the matched pairs)."""
        lens = torch.as_tensor(list(cap_lens), dtype=torch.int32)
        l0, l1, a2 = _GloriaLocalFn.apply(img, words, lens, float(self.temp1), float(self.temp2), float(self.temp3))
        B, ih, iw = img.shape[0], img.shape[2], img.shape[3]
        Tp = a2.shape[1] // B
        att_maps = [a2[i, i * Tp:i * Tp + int(cap_lens[i]), :ih * iw].reshape(1, int(cap_lens[i]), ih, iw) for i in range(B)]
        return l0, l1, att_maps
