"""ConVIRT / InfoNCE / GLoRIA losses on the HIP path (same constructors and return tuples as the reference).

ref: vilmedic/blocks/losses/selfsup/ConVIRTLoss.py:5-38, InfoNCELoss.py:5-24, GLoRIALoss.py:5-170.

The [B,B] similarity is one bf16 MFMA GEMM with fp32 output (alpha = 1/tau); row / column log-sum-exp, the diagonal
and the gradient matrix G are produced by the kernels in csrc/contrastive.hip; dA = G B / tau and dB = G^T A / tau are
two more GEMMs.  When torch.distributed is initialised with world_size > 1 the text / image embeddings are
all-gathered first (RCCL), so every rank sees the GLOBAL batch of negatives (SURVEY §8e -- a capability the reference
lacks: under DDP it contrasts within the local shard only, conVIRT.py:97-100).
"""
import os

import torch
import torch.nn as nn

from ... import ops
from ..._lib import check, lib, ptr, stream

BF16 = torch.bfloat16
FUSED = os.environ.get("VM_CONTRASTIVE_FUSED", "1") != "0"     # 0: the round-1 path (S materialised in fp32, three scalar passes) for A/B


def _pad8(n):
    return (n + 7) // 8 * 8


class _SimilarityLossFn(torch.autograd.Function):
    """(a [R,D], b [C,D]) -> per-row losses  row_i = lse_j S_ij - S_ii,  col_i = lse_j S_ji - S_ii  with
    S = n(a) n(b)^T * inv_tau  (n = L2 normalisation when ``normalize``)."""

    @staticmethod
    def forward(ctx, a, b, normalize, inv_tau, eps):
        R, D = a.shape
        Cn = b.shape[0]
        if D % 8:
            raise ValueError("embedding dim must be a multiple of 8")
        dev = a.device
        a32, b32 = a.detach().float().contiguous(), b.detach().float().contiguous()
        ah = torch.zeros(_pad8(R), D, dtype=BF16, device=dev)
        bh = torch.zeros(_pad8(Cn), D, dtype=BF16, device=dev)
        na = torch.empty(R, dtype=torch.float32, device=dev)
        nb = torch.empty(Cn, dtype=torch.float32, device=dev)
        L = lib()
        check(L.vm_rownorm_cast(ptr(a32), ptr(ah), ptr(na), R, D, int(normalize), eps, stream()), "vm_rownorm_cast")
        check(L.vm_rownorm_cast(ptr(b32), ptr(bh), ptr(nb), Cn, D, int(normalize), eps, stream()), "vm_rownorm_cast")
        lse_r = torch.empty(R, dtype=torch.float32, device=dev)
        lse_c = torch.empty(Cn, dtype=torch.float32, device=dev)
        diag = torch.empty(R, dtype=torch.float32, device=dev)
        n = min(R, Cn)
        if FUSED:       # S tile by tile on the MFMA, reduced in LDS: the [R, C] matrix never reaches HBM
            ws = torch.empty(L.vm_contrastive_ws(R, Cn), dtype=torch.uint8, device=dev)
            if n < R:
                diag.zero_()
            check(L.vm_contrastive_fwd(ptr(ah), ptr(bh), R, Cn, D, inv_tau, 0, ptr(lse_r), ptr(lse_c), ptr(diag), ptr(ws), ws.numel(), stream()),
                  "vm_contrastive_fwd")
            S, ldS = None, 0
        else:
            ldS = (Cn + 3) // 4 * 4
            S = torch.empty(R, ldS, dtype=torch.float32, device=dev)
            ops.gemm(ah, 0, bh, 0, S, R, Cn, D, alpha=inv_tau)
            check(L.vm_lse_rows_f32(ptr(S), ldS, ptr(lse_r), ptr(diag), R, Cn, 0, stream()), "vm_lse_rows_f32")
            check(L.vm_lse_cols_f32(ptr(S), ldS, ptr(lse_c), R, Cn, stream()), "vm_lse_cols_f32")
        ctx.save_for_backward(a32, b32, ah, bh, na, nb, S, lse_r, lse_c)
        ctx.meta = (normalize, inv_tau, eps, R, Cn, D, ldS)
        return lse_r[:n] - diag[:n], lse_c[:n] - diag[:n]

    @staticmethod
    def backward(ctx, g_row, g_col):
        a32, b32, ah, bh, na, nb, S, lse_r, lse_c = ctx.saved_tensors
        normalize, inv_tau, eps, R, Cn, D, ldS = ctx.meta
        dev = ah.device
        gr = torch.zeros(R, dtype=torch.float32, device=dev)
        gc = torch.zeros(Cn, dtype=torch.float32, device=dev)
        n = min(R, Cn)
        gr[:n] = g_row.float()
        gc[:n] = g_col.float()
        ldg = _pad8(Cn)
        G = torch.zeros(_pad8(R), ldg, dtype=BF16, device=dev)
        if S is None:   # fused: G from recomputed tiles
            check(lib().vm_contrastive_bwd(ptr(ah), ptr(bh), R, Cn, D, inv_tau, 0, ptr(lse_r), ptr(lse_c), ptr(gr), ptr(gc), ptr(G), ldg,
                                           stream()), "vm_contrastive_bwd")
        else:
            check(lib().vm_contrastive_grad(ptr(S), ldS, ptr(lse_r), ptr(lse_c), ptr(gr), ptr(gc), ptr(G), ldg, R, Cn, 0, stream()),
                  "vm_contrastive_grad")
        dah = torch.empty(R, D, dtype=torch.float32, device=dev)
        dbh = torch.empty(Cn, D, dtype=torch.float32, device=dev)
        ops.gemm(G, 0, bh, 1, dah, R, D, ldg, alpha=inv_tau)            # dA^ = G B^ / tau      (contraction over columns)
        ops.gemm(G, 1, ah, 1, dbh, Cn, D, _pad8(R), alpha=inv_tau)      # dB^ = G^T A^ / tau    (contraction over rows)
        if normalize:
            da = _normalize_bwd(a32, na, dah, eps)
            db = _normalize_bwd(b32, nb, dbh, eps)
        else:
            da, db = dah, dbh
        return da, db, None, None, None


def _normalize_bwd(x, norms, dxh, eps):
    """x^ = x / max(|x|, eps):  dx = (dx^ - x^ (x^ . dx^)) / |x|   for |x| > eps, dx^ / eps otherwise."""
    d = norms.clamp(min=eps)[:, None]
    xh = x / d
    proj = (xh * dxh).sum(1, keepdim=True)
    return torch.where(norms[:, None] > eps, (dxh - xh * proj) / d, dxh / d)


def _maybe_gather(*xs):
    """global negatives under data parallelism -- in TRAINING only: validation shards may differ by one batch between ranks, and a
    collective inside the loss would then wait forever; under no_grad the loss contrasts within the local shard (what the reference
    does under DDP)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and torch.is_grad_enabled():
        from ...parallel import all_gather_with_grad
        n = xs[0].shape[0]
        r = dist.get_rank()
        return [all_gather_with_grad(x.contiguous(), dist) for x in xs], slice(r * n, (r + 1) * n), dist.get_world_size()
    return list(xs), slice(None), 1


class ConVIRTLoss(nn.Module):
    def __init__(self, tau, lambda_, **kwargs):
        super().__init__()
        self.tau = tau
        self.lambda_ = lambda_

    def forward(self, linguistic, visual):
        (lg, vg), local, world = _maybe_gather(linguistic, visual)
        loss_l, loss_v = _SimilarityLossFn.apply(lg, vg, True, 1.0 / self.tau, 1e-8)
        loss = torch.mean(self.lambda_ * loss_v + (1 - self.lambda_) * loss_l)
        return loss, loss_l[local], loss_v[local]

    def __repr__(self):
        return "ConVIRTLoss(\n\t(cos_loss): CosineSimilarity()\n\t(tau): {}\n\t(lambda_): {}\n)".format(self.tau, self.lambda_)


class InfoNCELoss(nn.Module):
    """raw dot-product logits, CE both ways; ``tau`` is stored but NOT applied -- exactly as the reference."""

    def __init__(self, tau, **kwargs):
        super().__init__()
        self.tau = tau

    def forward(self, linguistic, visual):
        (lg, vg), local, world = _maybe_gather(linguistic, visual)
        loss_t, loss_i = _SimilarityLossFn.apply(lg, vg, False, 1.0, 1e-8)
        loss = ((loss_i + loss_t) / 2).mean()
        return loss, loss_t[local], loss_i[local]

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


class GLoRIALoss(nn.Module):
    def __init__(self, local_loss_weight=1.0, global_loss_weight=1.0, temp1=4.0, temp2=5.0, temp3=10.0):
        super().__init__()
        self.local_loss_weight, self.global_loss_weight = local_loss_weight, global_loss_weight
        self.temp1, self.temp2, self.temp3 = temp1, temp2, temp3

    def forward(self, global_features, local_features, word_embeddings, sent_embeddings, sents):
        # data parallel training: every caption is contrasted with the images of the GLOBAL batch (and vice versa), so the local
        # feature maps [b, D, 19, 19], the word embeddings [b, D, T] and both global embeddings are all-gathered (RCCL; backward =
        # sum over ranks + own slice) together with the word lists -- SURVEY §8e "GLoRIA local loss"
        (global_features, local_features, word_embeddings, sent_embeddings), _, world = _maybe_gather(
            global_features, local_features, word_embeddings, sent_embeddings)
        if world > 1:
            import torch.distributed as dist
            parts = [None] * world
            dist.all_gather_object(parts, list(sents))
            sents = [s_ for p_ in parts for s_ in p_]
        cap_lens = [len([w for w in sent if not w.startswith("[")]) + 1 for sent in sents]
        l0, l1, attn_maps = self._local(local_features.float(), word_embeddings.float(), cap_lens)
        # global: cosine-sim [B,B] * temp3 -> CE both ways == the HIP similarity loss with inv_tau = temp3 (mean over rows)
        row, col = _SimilarityLossFn.apply(global_features.float(), sent_embeddings.float(), True, self.temp3, 1e-8)
        loss = (l0 + l1) * self.local_loss_weight + (row.mean() + col.mean()) * self.global_loss_weight
        return loss, attn_maps

    def _local(self, img, words, cap_lens):
        """ref: GLoRIALoss.py:78-129.  The reference loops over captions, repeating each caption B times and attending the whole
        image batch (two bmm's with the feature dimension and a [B,D,T] weighted context per caption).  Here every
        (caption i, image j) pair is handled at once from ONE product, S[i,t,j,p] = <word_it, ctx_jp>:
          a1 = softmax over the caption's words t (ragged lengths masked), a2 = softmax over pixels p of temp1 * a1,
          <word_it, wctx_ijt> = sum_p a2 * S          (wctx = ctx_j a2 is never formed),
          |wctx_ijt|^2 = a2^T (ctx_j^T ctx_j) a2      (one [P,P] Gram matrix per image),
        so the feature dimension is contracted once, nothing is repeated, and captions of any length share the launch."""
        B, D = img.shape[0], img.shape[1]
        ih, iw = img.shape[2], img.shape[3]
        P = ih * iw
        T = int(max(cap_lens))
        ctx = img.reshape(B, D, P)
        w = words[:, :, :T]                                                        # [B,D,T]
        lens = torch.tensor(cap_lens, device=img.device)
        valid = torch.arange(T, device=img.device)[None, :] < lens[:, None]        # [B,T]
        S = (w.transpose(1, 2).reshape(B * T, D) @ ctx.permute(1, 0, 2).reshape(D, B * P)).view(B, T, B, P)
        a1 = torch.softmax(S.masked_fill(~valid[:, :, None, None], float("-inf")), dim=1)
        a2 = torch.softmax(a1 * self.temp1, dim=3)                                 # [B(cap),T,B(img),P]
        dot = (a2 * S).sum(3)                                                      # [B,T,B]
        gram = torch.bmm(ctx.transpose(1, 2), ctx)                                 # [B(img),P,P]
        a2j = a2.permute(2, 0, 1, 3).reshape(B, B * T, P)                          # image-major
        wn2 = (torch.bmm(a2j, gram) * a2j).sum(2).view(B, B, T).permute(1, 2, 0)   # |wctx|^2 as [B(cap),T,B(img)]
        wnorm = w.norm(dim=1)                                                      # [B,T]
        cos = dot / (wnorm[:, :, None] * wn2.clamp_min(1e-30).sqrt()).clamp(min=1e-8)
        row = torch.log((torch.exp(cos * self.temp2) * valid[:, :, None]).sum(1))  # [B(cap),B(img)]
        sims = row.t() * self.temp3                                                # [B(img),B(cap)]
        labels = torch.arange(B, device=img.device)
        att_maps = [a2[i, :cap_lens[i], i].reshape(1, cap_lens[i], ih, iw) for i in range(B)]
        return nn.functional.cross_entropy(sims, labels), nn.functional.cross_entropy(sims.t(), labels), att_maps
