"""BatchNorm with MICRO-BATCH statistics at full batch, on a hand-written channels-last kernel.

The reference's contrastive models run their towers in ``forward_batch_size`` micro-batches inside one autograd graph
(ref: vilmedic/models/selfsup/conVIRT.py:83-95 with forward_batch_size 4 in config/SELFSUP/convirt-mimic.yml:22; GLoRIA.py:92-105), so in
training mode every BatchNorm of the CNN normalises each micro-batch with ITS OWN statistics and updates the running statistics
micro-batch by micro-batch.  Executing that literally costs ``batch / forward_batch_size`` passes over a ~160-kernel CNN (64 passes of
4 images at the reference's ConVIRT setting: the step is launch-bound).  Here the CNN runs ONCE over the whole batch and only the
normalisation is grouped: per-(group, channel) mean / biased variance -> normalise, which is what the G sequential passes compute; the
running statistics receive the same G exponential-moving-average updates in closed form
(r_G = (1 - m)^G r_0 + m * sum_c (1 - m)^(G - 1 - c) s_c).  Convolutions, pooling and activations are per-sample, so nothing else
changes.  A trailing partial micro-batch (batch % g) is one more group of its own.

Round 5: on channels-last activations (what the MIOpen convolutions of the bf16 tower mode produce) the normalisation -- grouped or
not -- runs on ``vm_batchnorm_nhwc_fwd / _bwd`` (csrc/batchnorm.hip): statistics pass + normalise pass, with the residual add and the ReLU
that follow a BatchNorm in ResNet / DenseNet blocks inside the kernel (``forward(x, residual=..., relu=True)``, used by blocks/vision/cnn.py).
As a composition of torch reductions / elementwise kernels on an fp32 copy of the activation the grouping was ~130 ms of a 205 ms ConVIRT
step (profiles/r05_d_steady_kernel_stats_convirt.csv).  Tensors in the default NCHW layout keep the torch path below (plain torch
BatchNorm when nothing is grouped).  A channels-last input the kernel declines -- channel count not a multiple of 8 or above 2048, no affine
parameters, a residual of another dtype / layout -- takes the torch path too, and says so once per module (``log.warning``).
"""
import contextlib
import logging

import torch
import torch.nn as nn

from ... import ops
from ..._lib import VM_BF16, VM_F32, check, lib, ptr, stream

log = logging.getLogger(__name__)
_micro = {"size": 0}


@contextlib.contextmanager
def micro_batches(size):
    """inside: every MicroBatchNorm2d in training mode normalises groups of ``size`` consecutive samples separately"""
    old = _micro["size"]
    _micro["size"] = int(size or 0)
    try:
        yield
    finally:
        _micro["size"] = old


_ws = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = _ws[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return w


def _plain_running(bn):
    rm, rv = bn.running_mean, bn.running_var
    return rm is not None and rm.dtype == rv.dtype == torch.float32 and rm.is_cuda and rm.is_contiguous() and rv.is_contiguous()


def _grad_target(p):
    """the fp32 accumulation buffer a kernel may add this parameter's gradient onto directly: its ParamArena gradient view while that IS ``p.grad``
    (arena.py; what autograd's AccumulateGrad would add into with one more launch per parameter) -- else None: the gradient is returned to autograd"""
    if p is None:
        return None
    g, v = p.grad, getattr(p, "_vm_grad_view", None)
    if v is None or g is None or g.data_ptr() != v.data_ptr() or g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda:
        return None
    return g


def _affine_grads(gamma, beta, C, device):
    """-> (dgamma buffer, dbeta buffer, what backward returns for the two): the arena views (returns None, None) or fresh zeroed vectors"""
    tg, tb = _grad_target(gamma), _grad_target(beta)
    if gamma is not None and beta is not None and tg is not None and tb is not None:
        return tg, tb, None, None
    dgamma = torch.zeros(C, dtype=torch.float32, device=device) if gamma is not None else None
    dbeta = torch.zeros(C, dtype=torch.float32, device=device) if beta is not None else None
    return dgamma, dbeta, dgamma, dbeta


class _BatchNormNhwcFn(ops.Fn):
    """one vm_batchnorm_nhwc_fwd / _bwd pair over ``G`` groups of ``x.shape[0] // G`` images (channels-last x, residual, y)"""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, mean, rstd, G, relu, training, eps, running=None, eval_var=None):
        N, C, H, W = x.shape
        R = (N // G) * H * W
        y = torch.empty_like(x)
        dt = VM_BF16 if x.dtype == torch.bfloat16 else VM_F32
        var = eval_var                          # inference without a graph: rstd is None, the kernel takes rsqrt(running_var + eps) itself
        rm = rv = nbt = None
        mom = 0.0
        if training:
            mean = torch.empty(G, C, dtype=torch.float32, device=x.device)
            rstd = torch.empty(G, C, dtype=torch.float32, device=x.device)
            var = torch.empty(G, C, dtype=torch.float32, device=x.device)
            if running is not None:             # the moving averages are updated by the normalisation kernel (one update per group, in group order)
                rm, rv, nbt, mom = running
            ws = _workspace(lib().vm_batchnorm_nhwc_ws(G, R, C), x.device)
            check(lib().vm_batchnorm_nhwc_stats(ptr(x), C, None, 0, ptr(mean), ptr(rstd), ptr(var), C, ptr(nbt), G, R, C, eps, dt, ptr(ws), ws.numel(),
                                                stream()), "vm_batchnorm_nhwc_stats")
        check(lib().vm_batchnorm_nhwc_apply(ptr(x), C, ptr(residual), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(var), C, ptr(rm), ptr(rv),
                                            ptr(nbt), mom, eps, G, R, C, dt, int(relu), stream()), "vm_batchnorm_nhwc_apply")
        ctx.save_for_backward(x, residual, gamma, beta, mean, rstd)
        ctx.meta = (G, R, C, dt, relu, training)
        if training:
            ctx.mark_non_differentiable(mean, var)
            return y, mean, var
        return y, None, None

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, residual, gamma, beta, mean, rstd = ctx.saved_tensors
        G, R, C, dt, relu, training = ctx.meta
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        want_res = residual is not None and ctx.needs_input_grad[1]
        dres = torch.empty_like(x) if want_res else None
        dgamma, dbeta, ret_g, ret_b = _affine_grads(gamma, beta, C, x.device)
        ws = _workspace(lib().vm_batchnorm_nhwc_ws(G, R, C), x.device)
        check(lib().vm_batchnorm_nhwc_bwd(ptr(dy), ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dres), ptr(dgamma),
                                          ptr(dbeta), G, R, C, dt, int(relu), int(training), ptr(ws), ws.numel(), stream()), "vm_batchnorm_nhwc_bwd")
        return dx, dres, ret_g, ret_b, None, None, None, None, None, None, None, None


def _nhwc_ok(x, C):
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and C % 8 == 0 and C <= 2048 and x.shape[0] > 0
            and x.is_contiguous(memory_format=torch.channels_last))


class MicroBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state-dict keys) whose training statistics can be grouped per micro-batch and whose
    channels-last path is the HIP kernel.  ``forward(x, residual=None, relu=False)`` computes ``relu?(bn(x) + residual?)``."""

    def forward(self, x, residual=None, relu=False):
        C = self.num_features
        if _nhwc_ok(x, C) and self.affine and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype
                                                                      and residual.is_contiguous(memory_format=torch.channels_last))):
            return self._forward_hip(x, residual, relu)
        if x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() \
                and not getattr(self, "_vm_declined", False):
            self._vm_declined = True
            log.warning("MicroBatchNorm2d(%d): channels-last input declined by the HIP kernel (C %% 8 = %d, C <= 2048: %s, affine: %s, residual "
                        "matches: %s) -- this layer runs the torch path", C, C % 8, C <= 2048, self.affine,
                        residual is None or (residual.shape == x.shape and residual.dtype == x.dtype))
        y = self._forward_torch(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y

    # ------------------------------------------------------------------ channels-last: csrc/batchnorm.hip
    def _forward_hip(self, x, residual, relu):
        g = _micro["size"]
        B = x.shape[0]
        use_batch_stats = self.training or not self.track_running_stats
        if not use_batch_stats:
            if _plain_running(self) and not torch.is_grad_enabled():
                return _BatchNormNhwcFn.apply(x, residual, self.weight, self.bias, self.running_mean, None, 1, relu, False, self.eps, None, self.running_var)[0]
            mean = self.running_mean.float().view(1, -1).contiguous()
            rstd = torch.rsqrt(self.running_var.float() + self.eps).view(1, -1).contiguous()
            return _BatchNormNhwcFn.apply(x, residual, self.weight, self.bias, mean, rstd, 1, relu, False, self.eps)[0]
        grouped = self.training and 0 < g < B and self.track_running_stats
        G = B // g if grouped else 1
        main = G * g if grouped else B
        parts = [(0, main, G)] + ([(main, B, 1)] if main < B else [])          # the trailing partial micro-batch is a group of its own
        running = self._running_args(x.device)
        outs, stats = [], []
        for lo, hi, Gp in parts:
            xs = x if (lo == 0 and hi == B) else x[lo:hi]
            rs = None if residual is None else (residual if (lo == 0 and hi == B) else residual[lo:hi])
            y, mean, var = _BatchNormNhwcFn.apply(xs, rs, self.weight, self.bias, None, None, Gp, relu, True, self.eps, running)
            outs.append(y)
            stats.append((mean, var, (hi - lo) // Gp * x.shape[2] * x.shape[3]))
        if self.training and self.track_running_stats and running is None:
            self._update_running(stats)
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def _running_args(self, device):
        """(running_mean, running_var, num_batches_tracked, momentum or -1 for the cumulative average) when the kernel can update them in place"""
        if not (self.training and self.track_running_stats):
            return None
        rm, rv, nbt = self.running_mean, self.running_var, self.num_batches_tracked
        if not (rm.dtype == rv.dtype == torch.float32 and nbt.dtype == torch.int64 and rm.device == rv.device == nbt.device == device
                and rm.is_contiguous() and rv.is_contiguous()):
            return None
        return rm, rv, nbt, (-1.0 if self.momentum is None else float(self.momentum))

    @torch.no_grad()
    def _update_running(self, stats):
        """the exponential moving averages after one update per group, in closed form; ``momentum=None`` is nn.BatchNorm2d's cumulative
        average (factor 1 / num_batches_tracked per update): after Gt more updates r = (n0 r0 + sum_c s_c) / (n0 + Gt)"""
        means = torch.cat([m for m, _, _ in stats])
        uvars = torch.cat([v * (n / max(n - 1, 1)) for _, v, n in stats])         # running_var takes the unbiased estimate
        Gt = means.shape[0]
        if self.momentum is None:
            n0 = self.num_batches_tracked.to(torch.float32)
            self.running_mean.mul_(n0).add_(means.sum(0).to(self.running_mean.dtype)).div_(n0 + Gt)
            self.running_var.mul_(n0).add_(uvars.sum(0).to(self.running_var.dtype)).div_(n0 + Gt)
            self.num_batches_tracked += Gt
            return
        mom = self.momentum
        w = mom * (1.0 - mom) ** torch.arange(Gt - 1, -1, -1, device=means.device, dtype=torch.float32)      # weight of group c
        keep = (1.0 - mom) ** Gt
        self.running_mean.mul_(keep).add_((w[:, None] * means).sum(0).to(self.running_mean.dtype))
        self.running_var.mul_(keep).add_((w[:, None] * uvars).sum(0).to(self.running_var.dtype))
        self.num_batches_tracked += Gt

    # ------------------------------------------------------------------ NCHW (default layout): torch
    def _forward_torch(self, x):
        g = _micro["size"]
        B = x.shape[0]
        if not self.training or g <= 0 or g >= B or not self.track_running_stats:
            return super().forward(x)
        G = B // g
        main, rest = x[:G * g], x[G * g:]
        C = x.shape[1]
        xg = main.reshape(G, g, C, *x.shape[2:])
        red = (1, 3, 4)
        var, mean = torch.var_mean(xg.float(), dim=red, unbiased=False, keepdim=True)        # [G,1,C,1,1]
        y = (xg - mean.to(x.dtype)) * torch.rsqrt(var + self.eps).to(x.dtype)
        if self.affine:
            y = y * self.weight.view(1, 1, C, 1, 1) + self.bias.view(1, 1, C, 1, 1)
        y = y.reshape(main.shape)
        n = g * main[0, 0].numel()
        self._update_running([(mean.view(G, C), var.view(G, C), n)])
        if rest.shape[0]:
            y = torch.cat([y, super().forward(rest)])
        return y


# ----------------------------------------------------------------------------- DenseNet block on ONE feature buffer
class _DenseState:
    """what the layers of one dense-block call share: the channels-last feature buffer [N, Ctot, H, W], the batch statistics of its channels
    ([G, Ctot]; a channel's statistics are the same for every norm1 that reads it, so each layer computes them for its NEW channels only) and, during
    the backward pass, the gradient buffer every norm1 accumulates into"""
    __slots__ = ("buf", "mean", "rstd", "var", "G", "R", "Ctot", "dt", "grad", "eps")


class _DenseNormFn(ops.Fn):
    """append ``new`` (the block input or the previous layer's growth channels) to the buffer at channel ``lo`` and return relu(norm1(buffer[:, :hi]))"""

    @staticmethod
    def forward(ctx, new, gamma, beta, st, bn, lo, hi):
        buf = st.buf
        new = new.contiguous(memory_format=torch.channels_last)
        N, _, H, W = buf.shape
        h = torch.empty((N, hi, H, W), dtype=buf.dtype, device=buf.device, memory_format=torch.channels_last)
        esz = buf.element_size()
        training = bn.training
        if training:
            rm, rv, nbt, mom = bn._running_args(buf.device)
            ws = _workspace(lib().vm_batchnorm_nhwc_ws(st.G, st.R, hi - lo), buf.device)
            check(lib().vm_batchnorm_nhwc_stats(ptr(new), hi - lo, buf.data_ptr() + lo * esz, st.Ctot, st.mean.data_ptr() + 4 * lo, st.rstd.data_ptr() + 4 * lo,
                                                st.var.data_ptr() + 4 * lo, st.Ctot, ptr(nbt), st.G, st.R, hi - lo, st.eps, st.dt, ptr(ws), ws.numel(), stream()),
                  "vm_batchnorm_nhwc_stats")
            mean, rstd, ldm, G, R = st.mean, st.rstd, st.Ctot, st.G, st.R
            check(lib().vm_batchnorm_nhwc_apply(ptr(buf), st.Ctot, None, ptr(h), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(st.var), ldm, ptr(rm), ptr(rv),
                                                ptr(nbt), mom, st.eps, G, R, hi, st.dt, 1, stream()), "vm_batchnorm_nhwc_apply")
        else:
            buf[:, lo:hi].copy_(new)
            if _plain_running(bn) and not any(ctx.needs_input_grad[:3]):
                mean, rstd, var = bn.running_mean, None, bn.running_var        # rsqrt(running_var + eps) in the kernel
            else:
                mean = bn.running_mean.float().view(1, -1).contiguous()
                rstd = torch.rsqrt(bn.running_var.float() + bn.eps).view(1, -1).contiguous()
                var = None
            ldm, G, R = hi, 1, st.G * st.R
            check(lib().vm_batchnorm_nhwc_apply(ptr(buf), st.Ctot, None, ptr(h), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(var), ldm, None, None, None, 0.0,
                                                bn.eps, G, R, hi, st.dt, 1, stream()), "vm_batchnorm_nhwc_apply")
        ctx.st, ctx.stat, ctx.meta = st, (mean, rstd, gamma, beta), (ldm, G, R, lo, hi, training)
        return h

    @staticmethod
    def backward(ctx, dh):
        st = ctx.st
        mean, rstd, gamma, beta = ctx.stat
        ldm, G, R, lo, hi, training = ctx.meta
        grad = st.grad
        if grad is None:
            raise RuntimeError("dense block backward: the gradient buffer of the block is missing (the block output's gradient has not arrived)")
        dh = dh.contiguous(memory_format=torch.channels_last)
        if dh.dtype != grad.dtype:
            dh = dh.to(grad.dtype)
        dgamma, dbeta, ret_g, ret_b = _affine_grads(gamma, beta, hi, grad.device)
        ws = _workspace(lib().vm_batchnorm_nhwc_ws(G, R, hi), grad.device)
        check(lib().vm_batchnorm_nhwc_bwd_ex(ptr(dh), ptr(st.buf), st.Ctot, None, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ldm, ptr(grad), st.Ctot, 1, None,
                                             ptr(dgamma), ptr(dbeta), G, R, hi, st.dt, 1, int(training), ptr(ws), ws.numel(), stream()), "vm_batchnorm_nhwc_bwd")
        dnew = grad[:, lo:hi].contiguous(memory_format=torch.channels_last) if ctx.needs_input_grad[0] else None
        if lo == 0:
            st.grad = None                      # the first layer's backward is the block's last
        return dnew, ret_g, ret_b, None, None, None, None


class _DenseCloseFn(ops.Fn):
    """append the last layer's channels: the buffer IS the block's output.  Backward: its gradient becomes the block's gradient buffer."""

    @staticmethod
    def forward(ctx, new, st, lo):
        st.buf[:, lo:].copy_(new)
        ctx.st, ctx.lo = st, lo
        return st.buf

    @staticmethod
    def backward(ctx, dout):
        st = ctx.st
        if dout.dtype != st.buf.dtype:
            dout = dout.to(st.buf.dtype)
        st.grad = dout.clone(memory_format=torch.channels_last)        # owned: the layers accumulate into it in place
        return st.grad[:, ctx.lo:].contiguous(memory_format=torch.channels_last), None, None


def dense_block_ok(layers, x):
    """can this block run on one feature buffer?  channels-last device tensor, every norm1 a MicroBatchNorm2d the kernel takes, one eps / mode for all"""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous() and x.shape[0] > 0 and layers):
        return False
    c, first = x.shape[1], layers[0].norm1
    g, B = _micro["size"], x.shape[0]
    if first.training and first.track_running_stats and 0 < g < B and B % g:
        return False
    for layer in layers:
        bn = layer.norm1
        if not (isinstance(bn, MicroBatchNorm2d) and bn.affine and bn.track_running_stats and bn.num_features == c and c % 8 == 0 and c <= 2048
                and bn.eps == first.eps and bn.training == first.training and (not bn.training or bn._running_args(x.device) is not None)
                and isinstance(layer.relu1, nn.ReLU)):
            return False
        c += layer.conv2.out_channels
    return c % 8 == 0 and c <= 2048


def dense_block_forward(layers, x):
    """torchvision's _DenseBlock.forward (every layer reads the concatenation of the block input and all earlier layers' outputs) without the
    concatenations: the features live in one [N, Ctot, H, W] channels-last buffer, layer l normalises its first c_l channels in place of a
    torch.cat, and in the backward pass every norm1 adds its input gradient onto the same gradient buffer (no per-layer gradient adds)"""
    first = layers[0].norm1
    N, c0, H, W = x.shape
    st = _DenseState()
    st.Ctot = c0 + sum(layer.conv2.out_channels for layer in layers)
    st.buf = torch.empty((N, st.Ctot, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    g = _micro["size"]
    st.G = N // g if (first.training and 0 < g < N) else 1
    st.R = (N // st.G) * H * W
    st.dt = VM_BF16 if x.dtype == torch.bfloat16 else VM_F32
    st.eps, st.grad = first.eps, None
    if first.training:
        st.mean = torch.empty(st.G, st.Ctot, dtype=torch.float32, device=x.device)
        st.rstd = torch.empty_like(st.mean)
        st.var = torch.empty_like(st.mean)
    else:
        st.mean = st.rstd = st.var = None
    new, lo = x, 0
    for layer in layers:
        hi = lo + new.shape[1]
        h = _DenseNormFn.apply(new, layer.norm1.weight, layer.norm1.bias, st, layer.norm1, lo, hi)
        new = layer.conv2(bn_act(layer.norm2, layer.conv1(h), layer.relu2))
        if new.dtype != x.dtype:
            new = new.to(x.dtype)
        lo = hi
    return _DenseCloseFn.apply(new, st, lo)


def use_micro_batch_norm(module):
    """swap every nn.BatchNorm2d of ``module`` for a MicroBatchNorm2d sharing its parameters / buffers (state-dict keys unchanged)"""
    for name, child in list(module.named_children()):
        if type(child) is nn.BatchNorm2d:
            new = MicroBatchNorm2d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                   track_running_stats=child.track_running_stats)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            use_micro_batch_norm(child)
    return module


def bn_act(bn, x, relu=None, residual=None):
    """``relu(bn(x) + residual)`` -- inside ONE kernel when ``bn`` is a MicroBatchNorm2d on channels-last tensors and ``relu`` is a plain
    nn.ReLU (or None); the module-by-module composition otherwise"""
    if isinstance(bn, MicroBatchNorm2d) and (relu is None or isinstance(relu, nn.ReLU)):
        return bn(x, residual=residual, relu=relu is not None)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return relu(y) if relu is not None else y
