"""Decode step with KV cache + greedy / sampling / beam-search drivers on the HIP path.

Replaces HF ``GenerationMixin.generate`` as driven by the reference
(ref: vilmedic/blocks/huggingface/decoder/evaluation.py:73-78 -- greedy/beam for validation;
 vilmedic/blocks/rl/SCST.py:115-126,142-157 -- greedy and multinomial rollouts).
Token-selection semantics restate hf:generation/utils.py ``_sample`` (:2783-2975) and ``_beam_search`` (:3208-3520,
helpers :3008-3207): fp32 log-softmax, top-``2*beams`` over ``beams*V``, beam = idx // V, token = idx % V, the
early-stop heuristic of ``early_stopping=False``, pad-filling of finished rows.

MI355X-first differences (same results, different data flow):
  * cross-attention K|V of the image features are projected ONCE per layer and shared by all beams of a sample
    (HF repeats the encoder states per beam): the beams of a sample are simply extra query rows of one attention call;
  * a beam reorder never copies the self-attention KV cache (HF: hf:generation/utils.py:3478-3485): the attention
    kernel gathers keys through a small int32 row-index table, and reordering gathers that table;
  * K|V of the new token are written straight into the cache by the projection GEMM (strided output);
  * a decoder layer is 8 launches per step -- fused Q|K|V projection (LayerNorm of the previous sub-layer on load, Q to its buffer,
    K|V into the cache row), attention, output projection + residual, cross-Q (LayerNorm on load), cross-attention, output
    projection + residual, MLP up (LayerNorm on load, GELU), MLP down + residual (csrc/decode_gemm.hip; round 2: 12) -- and the
    ~100 launches of a step are captured into a HIP graph per position (the step is launch-bound: M = batch x beams rows); the decode
    state -- caches, cross K|V buffers, graphs -- is kept on the decoder and reused by later generate() calls of the same shape
    (VM_DECODE_GRAPH=0 turns both off).
"""
import ctypes as C_
import os
import weakref

import torch

from . import ops
from ._lib import VM_BF16, VM_F32, DecodeGemmArgs, check, lib, ptr, stream
from .arena import arena_of
from .nn import to_key_mask

BF16 = torch.bfloat16
F32 = torch.float32
DECODE_GRAPH = os.environ.get("VM_DECODE_GRAPH", "1") != "0"
# Arithmetic of the decode step.  "fp32" (default): nothing is rounded below fp32 (csrc/decode_f32.hip: fp32 master weights, exact
# f32 MFMA, fp32 caches) so greedy / beam token ids are those of the reference's fp32 path; "bf16": the training-precision
# kernels (half the weight bytes per step; a near-tie between the two best logits may resolve differently).  Per call:
# generate(..., decode_dtype="bf16"); SCST's rollouts ask for bf16 (they are sampled anyway).
DECODE_DTYPE = os.environ.get("VM_DECODE_DTYPE", "fp32")


def _ln(x, aff, eps):
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().vm_layernorm_fwd(ptr(x), ptr(aff.weight), ptr(aff.bias), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps, stream()),
          "vm_layernorm_fwd")
    return y


def _attn(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dh, key_mask=None, kv_index=None, kv_index_ld=0, scale=None):
    o = torch.empty(B * Lq, H * dh, dtype=BF16, device=q.device)
    stats = torch.empty(B * H * Lq * 2, dtype=torch.float32, device=q.device)
    check(lib().vm_attention_fwd(ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(o), H * dh, ptr(stats),
                                 ptr(key_mask) if key_mask is not None else None, B, H, Lq, Lk, dh, scale or dh ** -0.5, 0, 0.0, 0, None,
                                 ptr(kv_index) if kv_index is not None else None, kv_index_ld, stream()), "vm_attention_fwd")
    return o


def _gemm32(x, w, bias, out, M, N, K, *, ldc=None, act=0, residual=None):
    """out[M,N] = act(x[M,K] w[N,K]^T + bias) + residual, everything fp32 (vm_gemm_f32)"""
    check(lib().vm_gemm_f32(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(out), ldc if ldc is not None else out.stride(0), M, N, K,
                            ptr(bias) if bias is not None else None, act, ptr(residual) if residual is not None else None,
                            residual.stride(0) if residual is not None else 0, stream()), "vm_gemm_f32")
    return out


def _ln32(x, aff, eps):
    y = torch.empty_like(x)
    check(lib().vm_layernorm_f32(ptr(x), ptr(aff.weight), ptr(aff.bias), ptr(y), x.shape[0], x.shape[1], eps, stream()), "vm_layernorm_f32")
    return y


def _attn32(q, k, ldk, v, ldv, rows, H, Lk, dh, q_per_kv, key_mask=None, kv_index=None, kv_index_ld=0, scale=None):
    o = torch.empty(rows, H * dh, dtype=F32, device=q.device)
    check(lib().vm_attention_decode_f32(ptr(q), q.stride(0), ptr(k), ldk, ptr(v), ldv, ptr(o), H * dh,
                                        ptr(key_mask) if key_mask is not None else None,
                                        ptr(kv_index) if kv_index is not None else None, kv_index_ld,
                                        rows, H, Lk, dh, q_per_kv, scale or dh ** -0.5, stream()), "vm_attention_decode_f32")
    return o


class DecodeState:
    """Per-generate() state: cross K|V per layer, self K|V cache [rows*T, 2D] per layer, row-index table [rows, T]."""

    def __init__(self, decoder, enc, enc_mask, beams, max_length, dtype="bf16"):
        if dtype not in ("bf16", "fp32"):
            raise ValueError(f"decode_dtype must be 'bf16' or 'fp32', got {dtype!r}")
        self.f32 = dtype == "fp32"
        self.act = F32 if self.f32 else BF16
        self.dec = decoder
        cfg = decoder.config
        self.cfg = cfg
        self.arena = arena_of(decoder)
        self.arena.refresh()
        self.B = enc.shape[0]
        self.nb = beams
        self.M = self.B * beams
        self.T = max_length
        self.D = cfg.hidden_size
        self.H = cfg.num_attention_heads
        # head widths the attention kernels do not have (BertGenerationConfig's default 16 heads on hidden_size 768 = 48 columns,
        # ref:config/RRG/baseline-HF.yml): the step runs on zero-PADDED projection weights -- every head's Q / K / V rows and the matching
        # columns of the output projections are spread to the next kernel width (``Da`` = heads x padded width), rebuilt from the
        # parameters at every load_encoder().  Zero rows add nothing to QK^T, zero V columns give zero context columns that meet zero
        # weight columns; the softmax scale stays dh^-1/2.
        self.dh = self.D // self.H
        self.dhp = ops.padded_head_dim(self.dh)
        self.Da = self.H * self.dhp
        self.padded = self.dhp != self.dh
        self.scale = self.dh ** -0.5
        dev = enc.device
        self.S = enc.shape[1]
        self.layers = decoder.bert.encoder.layer
        # static buffers (their addresses are baked into the captured graphs)
        self.enc_mask = torch.ones(self.B, self.S, dtype=torch.uint8, device=dev) if enc_mask is not None else None
        self.cross_kv = [torch.empty(self.B * self.S, 2 * self.Da, dtype=self.act, device=dev) for _ in self.layers]
        self.self_kv = [torch.empty(self.M * self.T, 2 * self.Da, dtype=self.act, device=dev) for _ in self.layers]
        self.pw = None                    # per layer: the padded weights / biases of a padded-head step (built by load_encoder)
        self.index = torch.empty(self.M, self.T, dtype=torch.int32, device=dev)
        self.tok = torch.zeros(self.M, dtype=torch.long, device=dev)
        self.V = cfg.vocab_size
        self.emb_sh = self.arena.shadow_rows(decoder.bert.embeddings.word_embeddings.weight, decoder.padded_vocab)
        # the step's activations (static: their addresses are baked into the captured graphs): pre-LN running sum, its LayerNorm (the
        # residual), the attention queries, the MLP hidden
        ff = max(l.intermediate.dense.weight.shape[0] for l in self.layers)
        self.buf_s, self.buf_x = (torch.empty(self.M, self.D, dtype=self.act, device=dev) for _ in range(2))
        self.buf_q = torch.empty(self.M, max(self.D, self.Da), dtype=self.act, device=dev)
        self.buf_h = torch.empty(self.M, ff, dtype=self.act, device=dev)
        self.buf_stat = torch.empty(2 * self.M, dtype=torch.float32, device=dev)
        self.graphs = {}
        # BERT / RoBERTa decoders (blocks/huggingface/bert_models.py): the token-type row rides in a derived position table, and HF's
        # generate() numbers the positions 0, 1, 2, ... for them too -- it builds ``position_ids`` from the attention mask because their
        # forward accepts that argument (hf:generation/utils.py _prepare_position_ids_for_generation), which overrides RoBERTa's
        # pad-offset numbering of the training forward.  Reproduced as is: the reference decodes through that call.  Verified against
        # RobertaForCausalLM.generate / BertLMHeadModel.generate of the installed transformers (5.15): fixture G23, greedy and beam-4 token ids
        # bit-equal (tests/test_oracle_golden.py::test_g23_proto_decoder_loss_grads_and_decode_ids and its GPU twin); the pinned 4.55.3 is not
        # installable here (no network), so a 4.55.3 build whose RoBERTa generate() kept the pad-offset numbering would decode differently.
        self.pos_table = None
        if hasattr(decoder.bert.embeddings, "token_type_embeddings"):
            self.pos_table = torch.empty_like(decoder.bert.embeddings.position_embeddings.weight)
        self.load_encoder(enc, enc_mask)

    def load_encoder(self, enc, enc_mask):
        """(re)start a generation on this state: project the image features to every layer's cross K|V (once, shared by all
        beams of a sample), reset the row-index table.  Buffers keep their addresses, so captured graphs stay valid."""
        a = self.arena
        a.refresh()
        if self.pos_table is not None:
            e = self.dec.bert.embeddings
            with torch.no_grad():
                torch.add(e.position_embeddings.weight, e.token_type_embeddings.weight[0], out=self.pos_table)
        enc = enc.to(self.act).contiguous()
        enc2 = enc.view(self.B * self.S, enc.shape[2])
        if self.enc_mask is not None:
            self.enc_mask.copy_(to_key_mask(enc_mask))
        if self.padded:
            self._build_padded_weights()
        for li, (layer, kv) in enumerate(zip(self.layers, self.cross_kv)):
            ca = layer.crossattention.self
            if self.padded:
                w, bias = self.pw[li]["ckv"]
                if self.f32:
                    _gemm32(enc2, w, bias, kv, self.B * self.S, 2 * self.Da, enc2.shape[1])
                else:
                    ops.gemm(enc2, 0, w, 0, kv, self.B * self.S, 2 * self.Da, enc2.shape[1], bias=bias)
                continue
            if self.f32:
                _gemm32(enc2, a.f32_group([ca.key.weight, ca.value.weight]), a.f32_group([ca.key.bias, ca.value.bias]), kv,
                        self.B * self.S, 2 * self.D, enc2.shape[1])
                continue
            ops.gemm(enc2, 0, a.shadow_group([ca.key.weight, ca.value.weight]), 0, kv, self.B * self.S, 2 * self.D, enc2.shape[1],
                     bias=a.f32_group([ca.key.bias, ca.value.bias]))
        dev = enc.device
        self.index.copy_(torch.arange(self.M, device=dev, dtype=torch.int32)[:, None] * self.T
                         + torch.arange(self.T, device=dev, dtype=torch.int32)[None, :])

    @torch.no_grad()
    def _build_padded_weights(self):
        """zero-padded copies (step dtype; biases fp32) of the attention projections: rows / columns of head h at [h * dhp, h * dhp + dh)"""
        H, dh, w, dt = self.H, self.dh, self.dhp, self.act

        def rows(ws, bs):           # [n x D_out, K] stacked projections -> every head's rows spread to the padded width
            W = torch.cat([p.detach().float().view(H, dh, -1) for p in ws], 0)                       # [n*H, dh, K]
            Wp = torch.zeros(W.shape[0], w, W.shape[2], device=W.device)
            Wp[:, :dh] = W
            b = torch.cat([p.detach().float().view(H, dh) for p in bs], 0)
            bp = torch.zeros(b.shape[0], w, device=b.device)
            bp[:, :dh] = b
            return Wp.view(-1, W.shape[2]).to(dt).contiguous(), bp.view(-1).contiguous()

        def cols(p):                # [D, D_in] output projection -> its input columns spread likewise
            W = p.detach().float().view(p.shape[0], H, dh)
            Wp = torch.zeros(p.shape[0], H, w, device=W.device)
            Wp[:, :, :dh] = W
            return Wp.view(p.shape[0], -1).to(dt).contiguous()
        self.pw = []
        for layer in self.layers:
            sa, ca = layer.attention.self, layer.crossattention.self
            self.pw.append({"qkv": rows([sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias]),
                            "so": cols(layer.attention.output.dense.weight),
                            "cq": rows([ca.query.weight], [ca.query.bias]),
                            "ckv": rows([ca.key.weight, ca.value.weight], [ca.key.bias, ca.value.bias]),
                            "co": cols(layer.crossattention.output.dense.weight)})

    def reorder(self, parent_rows, upto):
        """beam j continues old beam parent_rows[j]: its history 0..upto-1 is the parent's (index-table gather, no cache copy)"""
        self.index[:, :upto] = self.index[parent_rows.long(), :upto]

    @torch.no_grad()
    def step(self, tokens, t):
        """tokens int64 [M] at position t  ->  fp32 logits [M, V] for position t+1 (valid until the next step(.., t))."""
        if not DECODE_GRAPH:
            return self._step(tokens, t)
        self.tok.copy_(tokens)
        entry = self.graphs.get(t)
        if entry is None:
            cur = torch.cuda.current_stream()
            warm = torch.cuda.Stream()
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):              # one eager pass first: lazy kernel attributes / allocator warm-up
                self._step(self.tok, t)
            cur.wait_stream(warm)
            graph = torch.cuda.CUDAGraph()
            with ops.capture(graph):
                out = self._step(self.tok, t)
            entry = self.graphs[t] = (graph, out)
        entry[0].replay()
        return entry[1]

    # ---- one decode step = embedding + 8 launches per layer + final LayerNorm + LM head (csrc/decode_gemm.hip: every LayerNorm but the
    # last is computed by the projection that consumes it, Q|K|V is one projection with two destinations)
    def _w(self, params):
        """the [N, K] operand of one weight or of an arena-adjacent group, in the step's dtype (fp32 masters / bf16 shadows)"""
        a = self.arena
        if len(params) == 1:
            return params[0] if self.f32 else a.shadow(params[0])
        return a.f32_group(params) if self.f32 else a.shadow_group(params)

    def _dg(self, A, W, C, M, N, K, *, bias=None, act=0, residual=None, ln=None, ln_out=None, c2=None, ldc2=0, split_n=0):
        """one vm_decode_gemm launch.  The kernel's domain is M <= 256 rows and K a multiple of 32 (bf16) / 16 (fp32), K <= 1024 with
        LayerNorm-on-load: more rows (64 samples x 8 beams, an SCST rollout of more than 128 samples) run as 256-row blocks, other
        widths through the general GEMMs with a separate LayerNorm."""
        step = 16 if self.f32 else 32
        if K % step or (ln is not None and K > 1024):
            return self._dg_general(A, W, C, M, N, K, bias=bias, act=act, residual=residual, ln=ln, ln_out=ln_out, c2=c2, ldc2=ldc2, split_n=split_n)
        if M > 256:
            for r0 in range(0, M, 256):
                rows = min(256, M - r0)
                self._dg(A[r0:], W, C[r0:], rows, N, K, bias=bias, act=act, residual=None if residual is None else residual[r0:], ln=ln,
                         ln_out=None if ln_out is None else ln_out[r0:], c2=None if c2 is None else c2[r0 * (ldc2 // c2.stride(0)):], ldc2=ldc2, split_n=split_n)
            return
        g = DecodeGemmArgs()
        g.dtype = VM_F32 if self.f32 else VM_BF16
        g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), C.data_ptr(), C.stride(0)
        g.M, g.N, g.K, g.act = M, N, K, act
        g.bias = bias.data_ptr() if bias is not None else None
        if residual is not None:
            g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
        if c2 is not None:
            g.c2, g.ldc2, g.split_n = c2.data_ptr(), ldc2, split_n
        if ln is not None:
            g.ln_gamma, g.ln_beta, g.ln_eps = ln.weight.data_ptr(), ln.bias.data_ptr(), self.cfg.layer_norm_eps
            if ln_out is not None:
                g.ln_out, g.ln_out_ld = ln_out.data_ptr(), ln_out.stride(0)
        check(lib().vm_decode_gemm(C_.byref(g), stream()), "vm_decode_gemm")

    def _dg_general(self, A, W, C, M, N, K, *, bias, act, residual, ln, ln_out, c2, ldc2, split_n):
        """the same projection through vm_gemm_bf16 / vm_gemm_f32 (any K that is a multiple of 8 / any K) + a separate LayerNorm"""
        if ln is not None:
            xn = ln_out if ln_out is not None else torch.empty(M, K, dtype=self.act, device=A.device)
            if self.f32:
                check(lib().vm_layernorm_f32(ptr(A), ptr(ln.weight), ptr(ln.bias), ptr(xn), M, K, self.cfg.layer_norm_eps, stream()), "vm_layernorm_f32")
            else:
                check(lib().vm_layernorm_fwd(ptr(A), ptr(ln.weight), ptr(ln.bias), ptr(xn), ptr(self.buf_stat), ptr(self.buf_stat[M:]), M, K,
                                             self.cfg.layer_norm_eps, stream()), "vm_layernorm_fwd")
            A = xn

        def run(Wp, bp, out, ldc, n):
            if self.f32:
                _gemm32(A, Wp, bp, out, M, n, K, ldc=ldc, act=act, residual=residual)
            else:
                ops.gemm(A, 0, Wp, 0, out, M, n, K, ldc=ldc, bias=bp, act=act, residual=residual)
        if c2 is None:
            return run(W, bias, C, C.stride(0), N)
        run(W, bias, C, C.stride(0), split_n)                       # Q -> C, K|V of the new token -> its cache row
        run(W[split_n:], None if bias is None else bias[split_n:], c2, ldc2, N - split_n)

    def _dgl(self, s, ln, x, W, C, M, N, K, **kw):
        """projection of LN(s), with x = LN(s) kept for the residual.  Up to 16 rows (fp32) / 32 rows (bf16) the LayerNorm rides on the
        projection's operand load (one launch); beyond, every one of the 144-192 workgroups recomputing the statistics of all rows costs
        more than the separate LayerNorm launch (measured per step, fused vs separate -- bf16: 16 rows 0.686 vs 0.751 ms, 32 rows 0.799 vs
        0.816, 64 rows 0.951 vs 0.904; fp32, with the single-read vm_layernorm_f32: 16 rows 1.032 vs 1.078, 64 rows 1.468 vs 1.34;
        256 rows 3.8 vs 2.1 ms)."""
        if M <= (16 if self.f32 else 32):
            return self._dg(s, W, C, M, N, K, ln=ln, ln_out=x, **kw)
        if self.f32:
            check(lib().vm_layernorm_f32(ptr(s), ptr(ln.weight), ptr(ln.bias), ptr(x), M, K, self.cfg.layer_norm_eps, stream()), "vm_layernorm_f32")
        else:
            check(lib().vm_layernorm_fwd(ptr(s), ptr(ln.weight), ptr(ln.bias), ptr(x), ptr(self.buf_stat), ptr(self.buf_stat[M:]), M, K,
                                         self.cfg.layer_norm_eps, stream()), "vm_layernorm_fwd")
        return self._dg(x, W, C, M, N, K, **kw)

    def _step(self, tokens, t):
        a, cfg, D, H, M, T = self.arena, self.cfg, self.D, self.H, self.M, self.T
        emb = self.dec.bert.embeddings
        s, x, q = self.buf_s, self.buf_x, self.buf_q
        fn = lib().vm_embedding_fwd_f32 if self.f32 else lib().vm_embedding_fwd
        pos = self.pos_table if self.pos_table is not None else emb.position_embeddings.weight
        check(fn(ptr(tokens.contiguous()), ptr(emb.word_embeddings.weight), ptr(pos), ptr(s), M, 1, D, t, stream()),
              "vm_embedding_fwd")
        ln = emb.LayerNorm                       # the LayerNorm that turns the running pre-LN sum ``s`` into the next sub-layer's input
        Da, dhp, sc = self.Da, self.dhp, self.scale
        for li, layer in enumerate(self.layers):
            sa = layer.attention.self
            cache = self.self_kv[li]
            if self.padded:                      # the same eight launches on the zero-padded projections (load_encoder built them)
                pw = self.pw[li]
                qp = q[:, :Da] if q.shape[1] == Da else q.as_strided((M, Da), (q.stride(0), 1))
                self._dgl(s, ln, x, pw["qkv"][0], qp, M, 3 * Da, D, bias=pw["qkv"][1], c2=cache[t:], ldc2=T * 2 * Da, split_n=Da)
                if self.f32:
                    ctx = _attn32(qp, cache, 2 * Da, cache[:, Da:], 2 * Da, M, H, t + 1, dhp, 1, kv_index=self.index, kv_index_ld=T, scale=sc)
                else:
                    ctx = _attn(qp, qp.stride(0), cache, 2 * Da, cache[:, Da:], 2 * Da, M, H, 1, t + 1, dhp, kv_index=self.index, kv_index_ld=T, scale=sc)
                blk = layer.attention.output
                self._dg(ctx, pw["so"], s, M, D, Da, bias=blk.dense.bias, residual=x)
                self._dgl(s, blk.LayerNorm, x, pw["cq"][0], qp, M, Da, D, bias=pw["cq"][1])
                kv = self.cross_kv[li]
                if self.f32:
                    ctx = _attn32(qp, kv, 2 * Da, kv[:, Da:], 2 * Da, M, H, self.S, dhp, self.nb, key_mask=self.enc_mask, scale=sc)
                else:
                    ctx = _attn(qp, qp.stride(0), kv, 2 * Da, kv[:, Da:], 2 * Da, self.B, H, self.nb, self.S, dhp, key_mask=self.enc_mask, scale=sc)
                blk = layer.crossattention.output
                self._dg(ctx, pw["co"], s, M, D, Da, bias=blk.dense.bias, residual=x)
                i, o = layer.intermediate.dense, layer.output.dense
                F = i.weight.shape[0]
                h = self.buf_h
                self._dgl(s, blk.LayerNorm, x, self._w([i.weight]), h, M, F, D, bias=i.bias, act=1)
                self._dg(h, self._w([o.weight]), s, M, D, F, bias=o.bias, residual=x)
                ln = layer.output.LayerNorm
                continue
            if getattr(sa, "fuse_q", False):     # Q|K|V in one launch: Q -> q, K|V of the new token -> cache row t
                self._dgl(s, ln, x, self._w([sa.query.weight, sa.key.weight, sa.value.weight]), q, M, 3 * D, D,
                          bias=a.f32_group([sa.query.bias, sa.key.bias, sa.value.bias]), c2=cache[t:], ldc2=T * 2 * D, split_n=D)
            else:
                self._dgl(s, ln, x, self._w([sa.query.weight]), q, M, D, D, bias=sa.query.bias)
                self._dg(x, self._w([sa.key.weight, sa.value.weight]), cache[t:].as_strided((M, 2 * D), (T * 2 * D, 1)), M, 2 * D, D,
                         bias=a.f32_group([sa.key.bias, sa.value.bias]))
            if self.f32:
                ctx = _attn32(q, cache, 2 * D, cache[:, D:], 2 * D, M, H, t + 1, D // H, 1, kv_index=self.index, kv_index_ld=T)
            else:
                ctx = _attn(q, D, cache, 2 * D, cache[:, D:], 2 * D, M, H, 1, t + 1, D // H, kv_index=self.index, kv_index_ld=T)
            blk = layer.attention.output
            self._dg(ctx, self._w([blk.dense.weight]), s, M, D, D, bias=blk.dense.bias, residual=x)
            ca = layer.crossattention.self
            self._dgl(s, blk.LayerNorm, x, self._w([ca.query.weight]), q, M, D, D, bias=ca.query.bias)
            kv = self.cross_kv[li]
            if self.f32:
                ctx = _attn32(q, kv, 2 * D, kv[:, D:], 2 * D, M, H, self.S, D // H, self.nb, key_mask=self.enc_mask)
            else:
                ctx = _attn(q, D, kv, 2 * D, kv[:, D:], 2 * D, self.B, H, self.nb, self.S, D // H, key_mask=self.enc_mask)
            blk = layer.crossattention.output
            self._dg(ctx, self._w([blk.dense.weight]), s, M, D, D, bias=blk.dense.bias, residual=x)
            i, o = layer.intermediate.dense, layer.output.dense
            F = i.weight.shape[0]
            h = self.buf_h
            self._dgl(s, blk.LayerNorm, x, self._w([i.weight]), h, M, F, D, bias=i.bias, act=1)
            self._dg(h, self._w([o.weight]), s, M, D, F, bias=o.bias, residual=x)
            ln = layer.output.LayerNorm
        hd, hl = self.dec.head_dense, self.dec.head_ln          # LM-head transform of BERT / RoBERTa decoders: dense -> GELU -> LayerNorm
        if self.f32:
            xl = _ln32(s, ln, cfg.layer_norm_eps)
            if hd is not None:
                self._dg(xl, self._w([hd.weight]), x, M, D, D, bias=hd.bias, act=1)
                xl = _ln32(x, hl, cfg.layer_norm_eps)
            V = self.V
            logits = torch.empty(M, (V + 3) // 4 * 4, dtype=F32, device=s.device)
            # (the decode GEMM's 64-row blocks: 128- / 256-row blocks measured 285 us for this product at 256 rows)
            self._dg(xl, emb.word_embeddings.weight, logits, M, V, D, bias=self.dec.lm_bias)
            return logits[:, :V]
        xl = _ln(s, ln, cfg.layer_norm_eps)
        if hd is not None:
            self._dg(xl, self._w([hd.weight]), x, M, D, D, bias=hd.bias, act=1)
            xl = _ln(x, hl, cfg.layer_norm_eps)
        return ops.lm_logits_f32(xl, self.emb_sh, self.dec.lm_bias, self.V)


class EnsembleState:
    """n-best checkpoint ensembling (ref: blocks/huggingface/decoder/beam_search.py:243-262,313-319, bin/ensemble.py:72-80):
    every model keeps its own encoder states and KV caches; the next-token logits are SUMMED before the log-softmax."""

    def __init__(self, decoders, encs, enc_masks, beams, max_length, dtype="bf16"):
        self.states, seen = [], set()
        for d, e, m in zip(decoders, encs, enc_masks):      # a decoder listed twice must not share one cached state
            fresh = id(d) in seen
            seen.add(id(d))
            self.states.append(DecodeState(d, e, m, beams, max_length, dtype) if fresh else _cached_state(d, e, m, beams, max_length, dtype))

    def reorder(self, parent_rows, upto):
        for st in self.states:
            st.reorder(parent_rows, upto)

    def step(self, tokens, t):
        logits = self.states[0].step(tokens, t)
        for st in self.states[1:]:
            logits = logits + st.step(tokens, t)
        return logits


_STATE_CACHE = weakref.WeakKeyDictionary()


def _cached_state(decoder, enc, enc_mask, beams, max_length, dtype="bf16"):
    """decode states (KV caches, cross K|V buffers, captured graphs) live on the decoder and are reused by generate()
    calls of the same shape; the key includes the arena's shadow buffer so a re-flattened model starts afresh"""
    if not DECODE_GRAPH:
        return DecodeState(decoder, enc, enc_mask, beams, max_length, dtype)
    cache = _STATE_CACHE.setdefault(decoder, {})        # weak: dies with the decoder; never part of deepcopy / pickling of the model
    arena = arena_of(decoder)
    key = (enc.shape[0], enc.shape[1], enc.shape[2], beams, max_length, enc_mask is not None, arena.shadow_flat.data_ptr(), str(enc.device), dtype)
    st = cache.get(key)
    if st is None:
        if len(cache) >= 4:
            cache.clear()
        st = cache[key] = DecodeState(decoder, enc, enc_mask, beams, max_length, dtype)
    else:
        st.load_encoder(enc, enc_mask)
    return st


def _decode_state(decoder, enc, enc_mask, beams, max_length, dtype=None):
    dtype = dtype or DECODE_DTYPE
    if isinstance(decoder, (list, tuple)):
        return EnsembleState(decoder, enc, enc_mask, beams, max_length, dtype), decoder[0], enc[0]
    return _cached_state(decoder, enc, enc_mask, beams, max_length, dtype), decoder, enc


def log_softmax_f32(logits):
    rows, V = logits.shape
    out = torch.empty(rows, V, dtype=torch.float32, device=logits.device)
    check(lib().vm_logsoftmax_f32(ptr(logits), logits.stride(0), ptr(out), rows, V, stream()), "vm_logsoftmax_f32")
    return out


def argmax_f32(x):
    rows, V = x.shape
    idx = torch.empty(rows, dtype=torch.long, device=x.device)
    check(lib().vm_argmax_f32(ptr(x), x.stride(0), ptr(idx), None, rows, V, stream()), "vm_argmax_f32")
    return idx


class GenerateOutput:
    def __init__(self, sequences, sequences_scores=None, scores=None):
        self.sequences, self.sequences_scores, self.scores = sequences, sequences_scores, scores


def _process(logits, bad_ids, top_k):
    """HF NoBadWordsLogitsProcessor (single-token bad words) then TopKLogitsWarper."""
    if bad_ids:
        logits[:, bad_ids] = -float("inf")
    if top_k:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]))[0][:, -1:]
        logits = logits.masked_fill(logits < kth, -float("inf"))
    return logits


_select_calls = [0]


@torch.no_grad()
def trim_to_last_eos(seq, eos_token_id):
    """HF's stopping point for a finished batch: generation ends right after the step at which its last row emitted eos"""
    is_eos = seq[:, 1:] == eos_token_id
    if seq.shape[1] > 1 and bool(is_eos.any(dim=1).all()):
        return seq[:, :int((is_eos.float().argmax(dim=1) + 1).max()) + 1]
    return seq


def sample(decoder, input_ids, enc, enc_mask, *, max_length, eos_token_id, pad_token_id, do_sample=False, top_k=None,
           bad_words_ids=None, output_scores=False, generator=None, decode_dtype=None, greedy_rows=None):
    """HF ``_sample``: greedy (argmax) or multinomial sampling, 1 sequence per batch row.  ``greedy_rows`` = n (with do_sample):
    the first n rows decode greedily on the RAW logits, the others sample from the processed ones -- two rollouts in one pass of
    launch-latency-bound decode steps (SCST: baseline + sampled rollout); callers split the rows and trim each part
    (trim_to_last_eos)."""
    B = input_ids.shape[0]
    st, decoder, enc0 = _decode_state(decoder, enc, enc_mask, 1, max_length, decode_dtype)
    dev = enc0.device
    seq = torch.full((B, max_length), pad_token_id, dtype=torch.long, device=dev)
    seq[:, 0] = input_ids[:, 0]
    bad = [w[0] for w in (bad_words_ids or [])]
    scores = []
    cur = 1
    fused = not output_scores and len(bad) <= 4 and (do_sample or not bad) and not (top_k and top_k > 256)
    if fused:
        # one selection launch per step (csrc/decode_select.hip): arg-max / filtered draw, finished-row padding, the write into ``seq``
        V = decoder.config.vocab_size
        g_rows = (int(greedy_rows) if greedy_rows else 0) if do_sample else B
        unf = torch.ones(B, dtype=torch.uint8, device=dev)
        nxt = torch.empty(B, dtype=torch.long, device=dev)
        ban = (C_.c_int32 * 4)(*(bad + [0] * (4 - len(bad)))) if bad else None
        # the draws follow torch's seeding (torch.manual_seed / an explicit CPU generator) through ONE host-side draw per call
        gen = generator if (generator is not None and generator.device.type == "cpu") else None
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=gen).item()) if do_sample else 0
        if generator is not None and generator.device.type != "cpu":
            seed = (int(generator.initial_seed()) * 0x9E3779B1 + _select_calls[0]) & (2 ** 63 - 1)
            _select_calls[0] += 1
        while cur < max_length:
            logits = st.step(seq[:, cur - 1], cur - 1)
            check(lib().vm_select_tokens(ptr(logits), logits.stride(0), B, V, g_rows, ban, len(bad), int(top_k or 0) if do_sample else 0, seed, ptr(nxt),
                                         ptr(seq), seq.stride(0), cur, ptr(unf), eos_token_id, pad_token_id, stream()), "vm_select_tokens")
            cur += 1
            if cur % 8 == 0 and not bool(unf.any()):           # host sync only every 8 steps; trimmed exactly below
                break
    # The one selection kernel covers every call the reference's shipped configs make (greedy; multinomial sampling with bad_words [[pad],[bos]]
    # and an optional top_k <= 256: ref:blocks/rl/SCST.py:142-157).  Outside its domain -- ``output_scores``, top_k > 256 (the YAML value is
    # free), more than 4 bad words, a greedy call with bad words -- the loop below serves the same contract step by step: HF's logits
    # processors on the fp32 logits (``_process``), then the library arg-max or ``torch.multinomial`` on the device (one launch more per step;
    # ``greedy_rows`` keeps its meaning: the first rows take the arg-max of the RAW logits).
    unfinished = torch.ones(B, dtype=torch.bool, device=dev)
    while not fused and cur < max_length:
        logits = st.step(seq[:, cur - 1], cur - 1)
        if do_sample and greedy_rows:
            g = int(greedy_rows)
            nxt = torch.empty(B, dtype=torch.long, device=dev)
            nxt[:g] = argmax_f32(logits[:g].contiguous())
            proc = _process(logits[g:].clone(), bad, top_k)
            nxt[g:] = torch.multinomial(torch.softmax(proc, dim=-1), 1, generator=generator).squeeze(1)
            if output_scores:
                logits = torch.cat([logits[:g], proc], dim=0)
        else:
            if do_sample or bad or output_scores:
                logits = _process(logits.clone(), bad, top_k if do_sample else None)
            if do_sample:
                nxt = torch.multinomial(torch.softmax(logits, dim=-1), 1, generator=generator).squeeze(1)
            else:
                nxt = argmax_f32(logits.contiguous())
        if output_scores:
            scores.append(logits)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_token_id))
        seq[:, cur] = nxt
        unfinished = unfinished & (nxt != eos_token_id)
        cur += 1
        if cur % 8 == 0 and not bool(unfinished.any()):      # host sync only every 8 steps; trimmed exactly below
            break
    # exact HF stopping point: generation ends right after the step at which the last row finished
    seq = seq[:, :cur]
    is_eos = seq[:, 1:] == eos_token_id
    if bool(is_eos.any(dim=1).all()):
        last = int((is_eos.float().argmax(dim=1) + 1).max()) + 1
        seq = seq[:, :last]
        scores = scores[:last - 1]
    return GenerateOutput(seq, None, tuple(scores) if output_scores else None)


def _gather(t, idx):
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.gather(t, 1, idx.expand(-1, -1, *t.shape[2:]))


@torch.no_grad()
def beam_search(decoder, input_ids, enc, enc_mask, *, max_length, eos_token_id, pad_token_id, num_beams, length_penalty=1.0,
                decode_dtype=None):
    """HF ``_beam_search`` (early_stopping=False, do_sample=False, num_return_sequences=1)."""
    B, nb, keep, prompt = input_ids.shape[0], num_beams, 2 * num_beams, 1
    st, decoder, enc0 = _decode_state(decoder, enc, enc_mask, nb, max_length, decode_dtype)
    dev = enc0.device
    V = decoder.config.vocab_size
    running = torch.full((B, nb, max_length), pad_token_id, dtype=torch.long, device=dev)
    running[:, :, 0] = input_ids[:, :1]
    seqs = running.clone()
    running_len = torch.zeros(B, nb, dtype=torch.long, device=dev)
    seq_len = torch.zeros(B, nb, dtype=torch.long, device=dev)
    running_scores = torch.zeros(B, nb, device=dev)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, nb), -1e9, device=dev)
    finished = torch.zeros(B, nb, dtype=torch.bool, device=dev)
    unsat = torch.ones(B, 1, dtype=torch.bool, device=dev)
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)]).to(dev)
    row_base = (torch.arange(B, device=dev) * nb)[:, None]
    topk_lp = torch.empty(B, keep, dtype=torch.float32, device=dev)
    topk_idx = torch.empty(B, keep, dtype=torch.long, device=dev)
    topk_ws = torch.empty(max(16, lib().vm_beam_topk_ws(B, min(nb, 8), keep)), dtype=torch.uint8, device=dev)
    cur = prompt
    while True:
        logits = st.step(running[:, :, cur - 1].reshape(-1), cur - 1)
        if nb <= 8:           # log-softmax + running scores + top-2nb: a per-row and a per-sample launch (csrc/loss.hip vm_beam_topk)
            check(lib().vm_beam_topk(ptr(logits), logits.stride(0), B, nb, V, ptr(running_scores), keep, ptr(topk_lp), ptr(topk_idx),
                                     ptr(topk_ws), topk_ws.numel(), stream()), "vm_beam_topk")
        else:
            logp = log_softmax_f32(logits.contiguous()).view(B, nb, V)
            logp = (logp + running_scores[:, :, None]).view(B, nb * V)
            topk_lp, topk_idx = torch.topk(logp, keep)
        src_beam = topk_idx // V
        tok = topk_idx % V
        cand = _gather(running, src_beam).clone()
        cand[:, :, cur] = tok
        cand_len = _gather(running_len, src_beam) + 1
        hits = (tok == eos_token_id) | (cur + 1 >= max_length)
        run_lp = topk_lp + hits.float() * -1e9
        nxt = torch.topk(run_lp, nb)[1]
        running, running_scores, running_len = _gather(cand, nxt), _gather(run_lp, nxt), _gather(cand_len, nxt)
        st.reorder((row_base + _gather(src_beam, nxt)).reshape(-1), cur)
        just = hits & top_mask[None, :]
        fin_lp = topk_lp / ((cur + 1 - prompt) ** length_penalty)
        fin_lp = fin_lp + (~unsat).float() * -1e9
        fin_lp = fin_lp + (~just).float() * -1e9
        m_seq, m_sc = torch.cat([seqs, cand], 1), torch.cat([beam_scores, fin_lp], 1)
        m_len, m_fin = torch.cat([seq_len, cand_len], 1), torch.cat([finished, just], 1)
        top = torch.topk(m_sc, nb)[1]
        seqs, beam_scores, seq_len, finished = _gather(m_seq, top), _gather(m_sc, top), _gather(m_len, top), _gather(m_fin, top)
        cur += 1
        best_possible = running_scores[:, :1] / ((cur - prompt) ** length_penalty)
        worst_fin = torch.where(finished, beam_scores.min(1, keepdim=True)[0], torch.full_like(beam_scores, -1e9))
        unsat = unsat & (best_possible > worst_fin).any(-1, keepdim=True)
        # HF stops as soon as no row can improve.  A step taken after that point changes nothing that is returned (a finished candidate is
        # only accepted while its row is unsatisfied: fin_lp gets -1e9 otherwise, and seqs[:, 0] / seq_len[:, 0] keep the best finished
        # hypothesis), so the host reads the flag every 4th step only -- a read per step left the ~40 selection launches of the next step
        # un-enqueued until the GPU had drained.  The last position always ends the loop.
        if cur + 1 > max_length or ((cur - prompt) % 4 == 0 and not bool(unsat.any())):
            break
    out_len = prompt + int(seq_len[:, 0].max())
    return GenerateOutput(seqs[:, 0, :out_len], beam_scores[:, 0])


def generate(decoder, input_ids=None, encoder_hidden_states=None, encoder_attention_mask=None, generation_config=None, **kw):
    """Front door with the keyword surface the reference uses (GenerationConfig kwargs or an object with those attributes)."""
    args = {}
    if generation_config is not None:
        src = generation_config if isinstance(generation_config, dict) else vars(generation_config)
        args.update({k: v for k, v in src.items() if not k.startswith("_")})
    args.update(kw)
    # ensemble call (the reference's keyword names, beam_search.py:226-227): hf_models = [decoder, ...] and
    # encoders_outputs = [{"encoder_hidden_states": ..., "encoder_attention_mask": ...}, ...]
    if args.get("hf_models"):
        decoder = list(args.pop("hf_models"))
        eo = args.pop("encoders_outputs")
        encoder_hidden_states = [e["encoder_hidden_states"] for e in eo]
        encoder_attention_mask = [e.get("encoder_attention_mask") for e in eo]
    cfg = (decoder[0] if isinstance(decoder, (list, tuple)) else decoder).config
    max_length = args.get("max_length") or 20
    eos = args.get("eos_token_id", cfg.eos_token_id)
    pad = args.get("pad_token_id", cfg.pad_token_id)
    nb = args.get("num_beams") or 1
    ret_dict = args.get("return_dict_in_generate", False)
    if input_ids is None:
        bos = args.get("bos_token_id", cfg.bos_token_id)
        e0 = encoder_hidden_states[0] if isinstance(encoder_hidden_states, (list, tuple)) else encoder_hidden_states
        input_ids = torch.full((e0.shape[0], 1), bos, dtype=torch.long, device=e0.device)
    if input_ids.shape[1] != 1:
        raise NotImplementedError("generate() starts from a single [bos] token (ref: evaluation.py:74)")
    if nb > 1:
        if args.get("do_sample"):
            raise NotImplementedError("beam sampling is outside the reference's call sites")
        out = beam_search(decoder, input_ids, encoder_hidden_states, encoder_attention_mask, max_length=max_length,
                          eos_token_id=eos, pad_token_id=pad, num_beams=nb, length_penalty=args.get("length_penalty", 1.0) or 1.0,
                          decode_dtype=args.get("decode_dtype"))
    else:
        out = sample(decoder, input_ids, encoder_hidden_states, encoder_attention_mask, max_length=max_length, eos_token_id=eos,
                     pad_token_id=pad, do_sample=bool(args.get("do_sample")), top_k=args.get("top_k"),
                     bad_words_ids=args.get("bad_words_ids"), output_scores=bool(args.get("output_scores")),
                     generator=args.get("generator"), decode_dtype=args.get("decode_dtype"), greedy_rows=args.get("greedy_rows"))
    return out if ret_dict else out.sequences
