"""MAPPO on the vectorised env: model, PPO loss/update, flat parameter bucket, data-parallel gradient exchange.

Mirrors the training side of the reference's pacman_mappo_resnet.py for the hot path only:
  MAPPOAgent            :97-211 (+ ResidualBlock :49-67, PositionalEncoding2D :69-95); parameter names are the
                        reference's, so its state_dict checkpoints load unchanged
  ppo_loss              :571-585 (per-minibatch advantage normalisation with the unbiased std, clipped surrogate,
                        0.5*MSE value loss, entropy bonus)
  PPOLearner.update_minibatch  :587-595 (zero_grad, backward, clip_grad_norm_ 0.5, Adam eps 1e-5, EMA 0.995)
  canonicalize_action   :232-238
Differences that are design, not semantics: parameters, gradients and the EMA copy live in ONE flat fp32 buffer each
(one RCCL all-reduce per optimizer step over xGMI, one fused Adam, one lerp for the EMA); the network runs under
bf16 autocast on the GPU (MFMA through MIOpen / hipBLASLt) with fp32 master weights; fp32 on CPU for the parity tests.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

GAMMA, GAE_LAMBDA = 0.99, 0.95                     # pacman_mappo_resnet.py:19-20
CLIP_EPS, VF_COEF, MAX_GRAD_NORM = 0.15, 0.5, 0.5  # :21-23
LR_START, LR_END = 2e-4, 4e-5                      # :27-28
ENT_COEF_START, ENT_COEF_END = 0.02, 0.004         # :29-30
EMA_DECAY = 0.995                                  # :38
SHAPING_SCALE = 0.1                                # :34


def layer_norm_small(x, ln):
    """nn.LayerNorm over a SMALL last dimension as var_mean + elementwise ops.  torch's native kernel launches one
    workgroup per row (RowwiseMomentsCUDAKernel): for the critic's [H*W*B, 32] token matrix that is 630 k tiny
    workgroups and 1.7 ms per call on MI355X (45 % of the optimizer step at batch 4096)."""
    xf = x.float()
    var, mean = torch.var_mean(xf, dim=-1, unbiased=False, keepdim=True)
    return ((xf - mean) * torch.rsqrt(var + ln.eps) * ln.weight + ln.bias).to(x.dtype)


_PENDING_ROWS = {}          # data_ptr of a partial-row gradient buffer -> (rows still to be added, floats per row, the buffer)
_DEFER_ROW_SUMS = [False]  # True only while PPOLearner._backward_group runs autograd: its gradient gather adds the rows; any other
                           # caller of these autograd functions gets the summed row from the backward call itself


class _row_sums_deferred:
    """Context for ONE backward call of the pmx_ffn / tok96 / tok32ln / *_tail families.  Those end with a small second-stage kernel
    that adds the partial rows of the parameter gradients into row 0 -- a launch on the critic's chain of the launch-bound
    512-sample step between every two backward kernels, although nothing before the gradient gather reads its result.  Under
    PPOLearner._backward_group the library skips that kernel and the buffer is remembered here; the gather (pmx_flatten_sum_to_f32)
    adds the rows while it copies.  `plain`: the parameter gradients are handed out as float32 views of row 0 (no cast copies)."""

    def __init__(self, lib, grad, floats, plain=True):
        self.lib, self.grad, self.floats = lib, grad, floats
        self.on = _DEFER_ROW_SUMS[0] and grad.is_cuda and plain

    def __enter__(self):
        if self.on:
            self.lib.pmx_defer_row_sums(1)
        return self

    def __exit__(self, *exc):
        if not self.on:
            return False
        self.lib.pmx_defer_row_sums(0)
        n = self.lib.pmx_last_partial_rows()
        if exc[0] is None and n > 0:
            _PENDING_ROWS[self.grad.data_ptr()] = (n, self.floats, self.grad)
        return False


def pending_rows_of(ptr):
    """(rows, floats per row) if the device address lies in row 0 of a buffer whose partial rows are still to be added, else (0, 0)."""
    for base, (n, floats, _) in _PENDING_ROWS.items():
        if base <= ptr < base + 4 * floats:
            return n, floats
    return 0, 0


def flush_pending_rows():
    """Adds whatever partial rows are still pending with the dedicated kernel (a consumer other than the learner's gather)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load() if _PENDING_ROWS else None
    for base, (n, floats, buf) in list(_PENDING_ROWS.items()):
        st = C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
        _lib.check(lib.pmx_sum_partial_rows(base, n, floats, st), "pmx_sum_partial_rows")
    _PENDING_ROWS.clear()


class _LN32Residual(torch.autograd.Function):
    """LayerNorm(x + a) over a 32-wide feature dimension through the fused HIP kernels pmx_ln32_forward/backward."""

    @staticmethod
    def forward(ctx, x, a, w, b, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x, a = x.contiguous(), a.contiguous()
        rows = x.numel() // 32
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        wf, bf = w.float().contiguous(), b.float().contiguous()
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.pmx_ln32_forward(x.data_ptr(), a.data_ptr(), wf.data_ptr(), bf.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), rows, float(eps), 0 if x.dtype == torch.float32 else 1, st), "pmx_ln32_forward")
        ctx.save_for_backward(x, a, wf, mean, rstd)
        ctx.wdtype = w.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x, a, wf, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        dz = torch.empty_like(x)
        partial = torch.empty(_lib.LN32_PARTIAL_ROWS, 64, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.pmx_ln32_backward(x.data_ptr(), a.data_ptr(), gy.data_ptr(), wf.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         dz.data_ptr(), partial.data_ptr(), x.numel() // 32,
                                         0 if x.dtype == torch.float32 else 1, st), "pmx_ln32_backward")
        dwb = partial.sum(0).to(ctx.wdtype)          # gradients carry the dtype of the parameters they belong to
        return dz, dz, dwb[:32], dwb[32:], None


def add_layer_norm_small(x, a, ln):
    """LayerNorm(x + a) for the critic tokens: the fused HIP kernel on the GPU (feature dimension 32, float32 or
    bfloat16), the reduce + elementwise formulation otherwise."""
    if x.is_cuda and x.shape[-1] == 32 and x.dtype == a.dtype and x.dtype in (torch.float32, torch.bfloat16):
        return _LN32Residual.apply(x, a, ln.weight, ln.bias, ln.eps)
    return layer_norm_small(x + a, ln)


class _FFNLayerNorm(torch.autograd.Function):
    """LayerNorm(x + linear2(relu(linear1(x)))) on [..., 32] bfloat16 tokens through pmx_ffn_forward / pmx_ffn_backward
    (csrc/pmx_critic.hip): one kernel each way, the 128-wide hidden activations never reach memory, backward recomputes."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, pack=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x = x.contiguous()
        dev = x.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if pack is None:
            pack = pack_ffn(w1, b1, w2, b2, gamma, beta)
        y = torch.empty_like(x)
        _lib.check(lib.pmx_ffn_forward(x.data_ptr(), pack.data_ptr(), y.data_ptr(), x.numel() // 32, float(eps), st), "pmx_ffn_forward")
        ctx.save_for_backward(x, pack)
        ctx.eps = float(eps)
        ctx.dtypes = [t.dtype for t in (w1, b1, w2, b2, gamma, beta)]
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x, pack = ctx.saved_tensors
        dev = x.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        dy = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty_like(x)
        grad = torch.empty((1 + _lib.GRAD_PARTIAL_ROWS) * _lib.FFN_GRAD_FLOATS, dtype=torch.float32, device=dev)   # row 0 = the result
        with _row_sums_deferred(lib, grad, _lib.FFN_GRAD_FLOATS, all(dt == torch.float32 for dt in ctx.dtypes)):
            _lib.check(lib.pmx_ffn_backward(x.data_ptr(), dy.data_ptr(), pack.data_ptr(), dx.data_ptr(), grad.data_ptr(), x.numel() // 32,
                                            ctx.eps, st), "pmx_ffn_backward")
        dw2, dw1 = grad[:4096].view(32, 128), grad[4096:8192].view(128, 32)
        db1, db2, dg, dbeta = grad[8192:8320], grad[8320:8352], grad[8352:8384], grad[8384:8416]
        outs = (dw1, db1, dw2, db2, dg, dbeta)
        return (dx,) + tuple(o.to(dt) for o, dt in zip(outs, ctx.dtypes)) + (None, None)


def pack_ffn(w1, b1, w2, b2, gamma, beta):
    """The feed-forward half's parameters in the kernels' operand layout (pmx_ffn_pack), on the current stream."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    ps = [t.detach().float().contiguous() for t in (w1, b1, w2, b2, gamma, beta)]
    pack = torch.empty(_lib.FFN_PACK_BYTES, dtype=torch.uint8, device=ps[0].device)
    st = C.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
    _lib.check(lib.pmx_ffn_pack(*[t.data_ptr() for t in ps], pack.data_ptr(), st), "pmx_ffn_pack")
    return pack


def pack_in_proj(w, b):
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
    pack = torch.empty(_lib.TOK96_PACK_BYTES, dtype=torch.uint8, device=wf.device)
    st = C.c_void_p(torch.cuda.current_stream(wf.device).cuda_stream)
    _lib.check(lib.pmx_tok96_pack(wf.data_ptr(), bf.data_ptr(), pack.data_ptr(), st), "pmx_tok96_pack")
    return pack


def pack_out_proj(w, b, gamma, beta):
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    ps = [t.detach().float().contiguous() for t in (w, b, gamma, beta)]
    pack = torch.empty(_lib.TOK32_PACK_BYTES, dtype=torch.uint8, device=ps[0].device)
    st = C.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
    _lib.check(lib.pmx_tok32ln_pack(*[t.data_ptr() for t in ps], pack.data_ptr(), st), "pmx_tok32ln_pack")
    return pack


def encoder_packs(layers):
    """The three parameter packs (in-projection, out-projection + norm1, feed-forward + norm2) of every given CriticEncoderLayer in
    ONE launch (pmx_encoder_pack) on the current stream -> [(pack_in, pack_out, pack_ffn), ...]."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(layers)
    arr = (_lib.EncoderLayerParams * n)()
    keep, out = [], []
    dev = layers[0].linear1.weight.device
    for i, l in enumerate(layers):
        mha = l.self_attn
        ps = [t.detach().float().contiguous() for t in (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                                                        l.norm1.weight, l.norm1.bias, l.linear1.weight, l.linear1.bias, l.linear2.weight,
                                                        l.linear2.bias, l.norm2.weight, l.norm2.bias)]
        keep.append(ps)
        packs = (torch.empty(_lib.TOK96_PACK_BYTES, dtype=torch.uint8, device=dev), torch.empty(_lib.TOK32_PACK_BYTES, dtype=torch.uint8, device=dev),
                 torch.empty(_lib.FFN_PACK_BYTES, dtype=torch.uint8, device=dev))
        for name, t in zip(("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "norm1_w", "norm1_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                            "norm2_w", "norm2_b"), ps):
            setattr(arr[i], name, t.data_ptr())
        arr[i].pack_in, arr[i].pack_out, arr[i].pack_ffn = (p.data_ptr() for p in packs)
        out.append(packs)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.pmx_encoder_pack(n, arr, st), "pmx_encoder_pack")
    return out


class _InProj96(torch.autograd.Function):
    """qkv = in_proj_weight a + in_proj_bias on [..., 32] bfloat16 tokens (pmx_tok96_forward / _backward)."""

    @staticmethod
    def forward(ctx, a, w, b, pack=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        a = a.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        if pack is None:
            pack = pack_in_proj(w, b)
        y = torch.empty(a.shape[:-1] + (96,), dtype=torch.bfloat16, device=a.device)
        _lib.check(lib.pmx_tok96_forward(a.data_ptr(), pack.data_ptr(), y.data_ptr(), a.numel() // 32, st), "pmx_tok96_forward")
        ctx.save_for_backward(a, pack)
        ctx.dtypes = (w.dtype, b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        a, pack = ctx.saved_tensors
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        dy = dy.to(torch.bfloat16).contiguous()
        da = torch.empty_like(a)
        grad = torch.empty((1 + _lib.GRAD_PARTIAL_ROWS) * _lib.TOK96_GRAD_FLOATS, dtype=torch.float32, device=a.device)
        with _row_sums_deferred(lib, grad, _lib.TOK96_GRAD_FLOATS, all(dt == torch.float32 for dt in ctx.dtypes)):
            _lib.check(lib.pmx_tok96_backward(a.data_ptr(), dy.data_ptr(), pack.data_ptr(), da.data_ptr(), grad.data_ptr(), a.numel() // 32, st),
                       "pmx_tok96_backward")
        return da, grad[:3072].view(96, 32).to(ctx.dtypes[0]), grad[3072:3168].to(ctx.dtypes[1]), None


class _InProj96Res(torch.autograd.Function):
    """_InProj96 that also hands its input back as a second output, `a_res`, for the layer's residual connection: the gradient of
    the residual branch then arrives HERE, and pmx_tok96_backward_res adds it to W^T dy inside the kernel (float32, rounded once)
    instead of autograd adding the two bfloat16 tensors with a launch of its own (one per encoder layer and optimizer step)."""

    @staticmethod
    def forward(ctx, a, w, b, pack=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        a = a.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        if pack is None:
            pack = pack_in_proj(w, b)
        y = torch.empty(a.shape[:-1] + (96,), dtype=torch.bfloat16, device=a.device)
        _lib.check(lib.pmx_tok96_forward(a.data_ptr(), pack.data_ptr(), y.data_ptr(), a.numel() // 32, st), "pmx_tok96_forward")
        ctx.save_for_backward(a, pack)
        ctx.dtypes = (w.dtype, b.dtype)
        return y, a.view_as(a)

    @staticmethod
    def backward(ctx, dy, dres):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        a, pack = ctx.saved_tensors
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        grad = torch.empty((1 + _lib.GRAD_PARTIAL_ROWS) * _lib.TOK96_GRAD_FLOATS, dtype=torch.float32, device=a.device)
        if dy is None:                                  # only the residual output was used
            grad.zero_()
            return dres, grad[:3072].view(96, 32).to(ctx.dtypes[0]), grad[3072:3168].to(ctx.dtypes[1]), None
        dy = dy.to(torch.bfloat16).contiguous()
        res = None if dres is None else dres.to(torch.bfloat16).contiguous()
        da = torch.empty_like(a)
        with _row_sums_deferred(lib, grad, _lib.TOK96_GRAD_FLOATS, all(dt == torch.float32 for dt in ctx.dtypes)):
            _lib.check(lib.pmx_tok96_backward_res(a.data_ptr(), dy.data_ptr(), pack.data_ptr(), None if res is None else res.data_ptr(),
                                                  da.data_ptr(), grad.data_ptr(), a.numel() // 32, st), "pmx_tok96_backward_res")
        return da, grad[:3072].view(96, 32).to(ctx.dtypes[0]), grad[3072:3168].to(ctx.dtypes[1]), None


class _OutProjAddLN(torch.autograd.Function):
    """LayerNorm(x + out_proj.weight a + out_proj.bias) on [..., 32] bfloat16 tokens (pmx_tok32ln_forward / _backward)."""

    @staticmethod
    def forward(ctx, x, a, w, b, gamma, beta, eps, pack=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x, a = x.contiguous(), a.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        if pack is None:
            pack = pack_out_proj(w, b, gamma, beta)
        y = torch.empty_like(x)
        _lib.check(lib.pmx_tok32ln_forward(x.data_ptr(), a.data_ptr(), pack.data_ptr(), y.data_ptr(), x.numel() // 32, float(eps), st),
                   "pmx_tok32ln_forward")
        ctx.save_for_backward(x, a, pack)
        ctx.eps = float(eps)
        ctx.dtypes = [t.dtype for t in (w, b, gamma, beta)]
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x, a, pack = ctx.saved_tensors
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        dy = dy.to(torch.bfloat16).contiguous()
        dx, da = torch.empty_like(x), torch.empty_like(a)
        grad = torch.empty((1 + _lib.GRAD_PARTIAL_ROWS) * _lib.TOK32_GRAD_FLOATS, dtype=torch.float32, device=a.device)
        with _row_sums_deferred(lib, grad, _lib.TOK32_GRAD_FLOATS, all(dt == torch.float32 for dt in ctx.dtypes)):
            _lib.check(lib.pmx_tok32ln_backward(x.data_ptr(), a.data_ptr(), dy.data_ptr(), pack.data_ptr(), dx.data_ptr(), da.data_ptr(),
                                                grad.data_ptr(), x.numel() // 32, ctx.eps, st), "pmx_tok32ln_backward")
        outs = (grad[:1024].view(32, 32), grad[1024:1056], grad[1056:1088], grad[1088:1120])
        return (dx, da) + tuple(o.to(dt) for o, dt in zip(outs, ctx.dtypes)) + (None, None)


def _fused_tokens_ok(x, *weights):
    return (MAPPOAgent.fused_ffn and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == 32
            and all(w.dtype == torch.float32 for w in weights))


def in_proj96(x, mha, pack=None):
    if _fused_tokens_ok(x, mha.in_proj_weight) and tuple(mha.in_proj_weight.shape) == (96, 32):
        return _InProj96.apply(x, mha.in_proj_weight, mha.in_proj_bias, pack)
    return token_linear(x, mha.in_proj_weight, mha.in_proj_bias)


def in_proj96_res(x, mha, pack=None):
    """-> (qkv, x_res): x_res is x, to be used for the layer's residual connection (see _InProj96Res)."""
    if _fused_tokens_ok(x, mha.in_proj_weight) and tuple(mha.in_proj_weight.shape) == (96, 32):
        return _InProj96Res.apply(x, mha.in_proj_weight, mha.in_proj_bias, pack)
    return token_linear(x, mha.in_proj_weight, mha.in_proj_bias), x


def out_proj_add_layer_norm(x, a, proj, ln, pack=None):
    if _fused_tokens_ok(x, proj.weight) and a.dtype == torch.bfloat16 and tuple(proj.weight.shape) == (32, 32):
        return _OutProjAddLN.apply(x, a, proj.weight, proj.bias, ln.weight, ln.bias, ln.eps, pack)
    return add_layer_norm_small(x, token_linear(a, proj.weight, proj.bias), ln)


def ffn_layer_norm(x, lin1, lin2, ln, pack=None):
    """The feed-forward half of the post-LN encoder layer: the fused HIP kernels for bf16 tokens of width 32 with a 128-wide
    hidden layer on the GPU, the separate ops otherwise."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == 32 and tuple(lin1.weight.shape) == (128, 32)
            and tuple(lin2.weight.shape) == (32, 128) and lin1.weight.dtype == torch.float32 and MAPPOAgent.fused_ffn):
        return _FFNLayerNorm.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ln.weight, ln.bias, ln.eps, pack)
    f = token_linear(F.relu(token_linear(x, lin1.weight, lin1.bias)), lin2.weight, lin2.bias)
    return add_layer_norm_small(x, f, ln)


def attention8_forward(qkv, want_lse=False, batch_major=False):
    """softmax(q k^T / sqrt(8)) v for 4 heads of 8 on the matrix cores (pmx_attn8_forward_layout): qkv [S, B, 96] bfloat16 ->
    [S, B, 32] bfloat16 (+ log-sum-exp [B, 4, S] float32); with batch_major qkv is [B, S, 96] and the result [B, S, 32]."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    (B, S, _) = qkv.shape if batch_major else (qkv.shape[1], qkv.shape[0], 0)
    qkv = qkv.contiguous()
    out = torch.empty(qkv.shape[0], qkv.shape[1], 32, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, 4, S, dtype=torch.float32, device=qkv.device) if want_lse else None
    st = C.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
    _lib.check(lib.pmx_attn8_forward_layout(qkv.data_ptr(), out.data_ptr(), lse.data_ptr() if want_lse else None, S, B,
                                            1 if batch_major else 0, st), "pmx_attn8_forward")
    return (out, lse) if want_lse else out


class _Attention8(torch.autograd.Function):
    """Differentiable wrapper of the MFMA attention kernels (pmx_attn8_forward_layout / pmx_attn8_backward_layout)."""

    @staticmethod
    def forward(ctx, qkv, batch_major):
        out, lse = attention8_forward(qkv, want_lse=True, batch_major=batch_major)
        ctx.save_for_backward(qkv.contiguous(), out, lse)
        ctx.batch_major = batch_major
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        qkv, out, lse = ctx.saved_tensors
        B, S = lse.shape[0], lse.shape[2]
        gout = gout.contiguous().to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        st = C.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
        _lib.check(lib.pmx_attn8_backward_layout(qkv.data_ptr(), out.data_ptr(), gout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), S, B,
                                                 1 if ctx.batch_major else 0, st), "pmx_attn8_backward")
        return dqkv, None


def attention8(qkv, batch_major=False):
    return _Attention8.apply(qkv, batch_major)


def pack_projector(w, b):
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
    pack = torch.empty(_lib.PROJ_PACK_BYTES, dtype=torch.uint8, device=wf.device)
    st = C.c_void_p(torch.cuda.current_stream(wf.device).cuda_stream)
    _lib.check(lib.pmx_proj_pack(wf.data_ptr(), bf.data_ptr(), pack.data_ptr(), st), "pmx_proj_pack")
    return pack


class _Projector(torch.autograd.Function):
    """tokens [B, H*W, 32] bf16 = conv3x3(8 -> 32)(obs) + bias + positional table, batch-major, through pmx_proj_forward / _backward
    (csrc/pmx_actor.hip): the observation planes are read as they are (bytes), no cast, no layout copy, no library convolution."""

    @staticmethod
    def forward(ctx, obs, w, b, pe, pack=None):
        import ctypes as C
        from . import _lib
        from .actor_tower import _OBS_CODE
        lib = _lib.load()
        obs = obs.contiguous()
        B, _, H, W = obs.shape
        if pack is None:
            pack = pack_projector(w, b)
        tok = torch.empty(B, H * W, 32, dtype=torch.bfloat16, device=obs.device)
        st = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        _lib.check(lib.pmx_proj_forward(obs.data_ptr(), _OBS_CODE[obs.dtype], pack.data_ptr(), pe.data_ptr(), tok.data_ptr(), B, H, W, st),
                   "pmx_proj_forward")
        ctx.save_for_backward(obs)
        ctx.dtypes = (w.dtype, b.dtype)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        import ctypes as C
        from . import _lib
        from .actor_tower import _OBS_CODE
        lib = _lib.load()
        (obs,) = ctx.saved_tensors
        B, _, H, W = obs.shape
        dev = obs.device
        dtok = dtok.to(torch.bfloat16).contiguous()
        part = torch.empty(_lib.PROJ_PARTIAL_ROWS * _lib.PROJ_GRAD_ROW_FLOATS, dtype=torch.float32, device=dev)
        dw = torch.empty(32, 8, 3, 3, dtype=torch.float32, device=dev)
        db = torch.empty(32, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pmx_proj_backward(obs.data_ptr(), _OBS_CODE[obs.dtype], dtok.data_ptr(), part.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                         B, H, W, st), "pmx_proj_backward")
        return None, dw.to(ctx.dtypes[0]), db.to(ctx.dtypes[1]), None, None


class _ActorTail(torch.autograd.Function):
    """logits = Linear_5(gelu(LayerNorm_512(h))) through pmx_actor_tail_forward / _backward (csrc/pmx_heads.hip): one launch each
    way (+ a row sum) instead of LayerNorm, GELU, a skinny GEMM, their backward kernels, two bias reductions and the casts."""

    @staticmethod
    def forward(ctx, h, lnw, lnb, w2, b2, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        h = h.contiguous()
        B = h.shape[0]
        logits = torch.empty(B, 5, dtype=torch.float32, device=h.device)
        stats = torch.empty(B, 2, dtype=torch.float32, device=h.device)
        st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        _lib.check(lib.pmx_actor_tail_forward(h.data_ptr(), 1 if h.dtype == torch.bfloat16 else 0, lnw.data_ptr(), lnb.data_ptr(), w2.data_ptr(),
                                              b2.data_ptr(), logits.data_ptr(), stats.data_ptr(), B, float(eps), st), "pmx_actor_tail_forward")
        ctx.save_for_backward(h, stats, lnw, lnb, w2)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        h, stats, lnw, lnb, w2 = ctx.saved_tensors
        B = h.shape[0]
        dlogits = dlogits.float().contiguous()
        dh = torch.empty_like(h)
        G = _lib.ACTOR_TAIL_GRAD_FLOATS
        grad = torch.empty((1 + _lib.HEADS_PARTIAL_ROWS) * G, dtype=torch.float32, device=h.device)
        st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        with _row_sums_deferred(lib, grad, G):
            _lib.check(lib.pmx_actor_tail_backward(h.data_ptr(), 1 if h.dtype == torch.bfloat16 else 0, stats.data_ptr(), dlogits.data_ptr(),
                                                   lnw.data_ptr(), lnb.data_ptr(), w2.data_ptr(), dh.data_ptr(), grad.data_ptr(), B, st), "pmx_actor_tail_backward")
        return dh, grad[2568:3080], grad[3080:3592], grad[:2560].view(5, 512), grad[2560:2565], None


class _CriticTail(torch.autograd.Function):
    """value = Linear_1(gelu(Linear_512(mean over tokens))) through pmx_critic_tail_forward / _backward: the mean pool, both
    linears, the GELU and -- backwards -- the broadcast of the pooled gradient over the tokens and all four parameter gradients."""

    @staticmethod
    def forward(ctx, tokens, w1, b1, w2, b2):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        tokens = tokens.contiguous()
        B, S, _ = tokens.shape
        value = torch.empty(B, dtype=torch.float32, device=tokens.device)
        pooled = torch.empty(B, 32, dtype=torch.float32, device=tokens.device)
        st = C.c_void_p(torch.cuda.current_stream(tokens.device).cuda_stream)
        _lib.check(lib.pmx_critic_tail_forward(tokens.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), value.data_ptr(),
                                               pooled.data_ptr(), B, S, st), "pmx_critic_tail_forward")
        ctx.save_for_backward(pooled, w1, b1, w2)
        ctx.S = S
        return value

    @staticmethod
    def backward(ctx, dvalue):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        pooled, w1, b1, w2 = ctx.saved_tensors
        B, S, dev = pooled.shape[0], ctx.S, pooled.device
        dvalue = dvalue.float().contiguous()
        dtok = torch.empty(B, S, 32, dtype=torch.bfloat16, device=dev)
        scratch = torch.empty(2 * B * 512, dtype=torch.bfloat16, device=dev)
        grad = torch.empty((1 + _lib.HEADS_PARTIAL_ROWS) * _lib.CRITIC_TAIL_GRAD_FLOATS, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with _row_sums_deferred(lib, grad, _lib.CRITIC_TAIL_GRAD_FLOATS):
            _lib.check(lib.pmx_critic_tail_backward(pooled.data_ptr(), dvalue.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), dtok.data_ptr(),
                                                    scratch.data_ptr(), grad.data_ptr(), B, S, st), "pmx_critic_tail_backward")
        return dtok, grad[:16384].view(512, 32), grad[16384:16896], grad[16896:17408].view(1, 512), grad[17408:17409]


def column_sums(t):
    """t.sum over all but the last dimension, in float32: the HIP column-sum kernel for a contiguous bfloat16 tensor on the
    GPU whose last dimension is a multiple of 8 (<= 256), torch's reduction otherwise."""
    C_ = t.shape[-1]
    if t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and C_ % 8 == 0 and 8 <= C_ <= 256 and t.numel() >= C_ * 4096:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        partial = torch.empty(_lib.COLSUM_BLOCKS, C_, dtype=torch.float32, device=t.device)
        st = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        _lib.check(lib.pmx_colsum_bf16(t.data_ptr(), t.numel() // C_, C_, partial.data_ptr(), st), "pmx_colsum_bf16")
        return partial.sum(0)
    if t.dim() == 1:
        return t.float()
    return t.sum(dim=tuple(range(t.dim() - 1)), dtype=torch.float32)


_DEBUG_KEEP = None    # tools/graph_debug.py: keeps (grad_out, bias_grad, weight_grad) of every token_linear backward alive


class _TokenLinear(torch.autograd.Function):
    """F.linear on a token tensor [S, B, in] with the weight gradient computed as S batched GEMMs of depth B followed by
    a sum over S.  hipBLASLt's choice for the flat [S*B, in]^T x [S*B, out] product (K = 630 k rows, a 32 x 128 result)
    is a 16x32x512 tile that takes 0.4-0.6 ms per call on MI355X; eight of them were 12 % of the optimizer step."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy.matmul(weight.to(gy.dtype)) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(gy.transpose(1, 2), x.to(gy.dtype)).sum(0).to(weight.dtype) if ctx.needs_input_grad[1] else None
        gb = column_sums(gy).to(weight.dtype) if ctx.needs_input_grad[2] else None
        if _DEBUG_KEEP is not None:
            _DEBUG_KEEP.append((gy, gb, gw))
        return gx, gw, gb


def token_linear(x, weight, bias):
    if x.dim() == 3 and torch.is_grad_enabled() and weight.requires_grad:
        if x.is_cuda and torch.is_autocast_enabled():
            dt = torch.get_autocast_gpu_dtype()
            x, weight, bias = x.to(dt), weight.to(dt), bias.to(dt)
        return _TokenLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class CriticEncoderLayer(nn.TransformerEncoderLayer):
    """The post-LN encoder layer of nn.TransformerEncoderLayer (norm_first=False, ReLU, dropout 0) with the same
    parameters and the same math, written out so that (a) the two LayerNorms over d_model = 32 use layer_norm_small and
    (b) the four projections use token_linear.  Self-attention follows nn.MultiheadAttention: packed in-projection,
    heads split as [B*h, S, d], scaled_dot_product_attention, out-projection."""

    def forward(self, src, src_mask=None, src_key_padding_mask=None, is_causal=False):
        x = src                                                       # [S, B, E]
        S, B, E = x.shape
        mha = self.self_attn
        h, d = mha.num_heads, E // mha.num_heads
        if x.is_cuda and torch.is_autocast_enabled() and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        qkv = in_proj96(x, mha) if (E == 32 and h == 4) else token_linear(x, mha.in_proj_weight, mha.in_proj_bias)
        if self.fused_ok(qkv, S):
            a = attention8(qkv) if torch.is_grad_enabled() else attention8_forward(qkv)   # hand-written MFMA attention
            x = out_proj_add_layer_norm(x, a, mha.out_proj, self.norm1)
            return ffn_layer_norm(x, self.linear1, self.linear2, self.norm2)
        q, k, v = qkv.chunk(3, dim=-1)
        q, k, v = (t.reshape(S, B * h, d).transpose(0, 1).reshape(B, h, S, d) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v)                    # [B, h, S, d]
        a = a.permute(2, 0, 1, 3).reshape(S, B, E)
        a = token_linear(a, mha.out_proj.weight, mha.out_proj.bias)
        x = add_layer_norm_small(x, a, self.norm1)
        f = token_linear(F.relu(token_linear(x, self.linear1.weight, self.linear1.bias)), self.linear2.weight, self.linear2.bias)
        return add_layer_norm_small(x, f, self.norm2)

    def fused_ok(self, t, S):
        """True when the hand-written kernels take this layer's tensors: bfloat16 on the GPU, embed 32, 4 heads, S in range."""
        mha = self.self_attn
        return (t.is_cuda and t.dtype == torch.bfloat16 and mha.embed_dim == 32 and mha.num_heads == 4
                and S <= (640 if torch.is_grad_enabled() else 1024))

    def packs(self):
        """This layer's three parameter packs (in-projection, out-projection + norm1, feed-forward + norm2) on the current stream."""
        mha = self.self_attn
        return (pack_in_proj(mha.in_proj_weight, mha.in_proj_bias),
                pack_out_proj(mha.out_proj.weight, mha.out_proj.bias, self.norm1.weight, self.norm1.bias),
                pack_ffn(self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm2.weight, self.norm2.bias))

    fold_residual_gradient = True   # development switch: False = autograd adds the in-projection's and the residual's gradients itself

    def forward_batch_major(self, x, packs=None):
        """The same layer on x [B, S, 32] bfloat16 (tokens of a sample contiguous: the memory order of a channels-last
        convolution output), hand-written kernels only; the caller has checked fused_ok.  packs: this layer's packs() made ahead."""
        mha = self.self_attn
        p_in, p_out, p_ffn = packs if packs is not None else (None, None, None)
        if torch.is_grad_enabled() and self.fold_residual_gradient:
            qkv, x = in_proj96_res(x, mha, p_in)      # the residual branch's gradient is added inside the in-projection's backward kernel
        else:
            qkv = in_proj96(x, mha, p_in)
        a = attention8(qkv, True) if torch.is_grad_enabled() else attention8_forward(qkv, batch_major=True)
        x = out_proj_add_layer_norm(x, a, mha.out_proj, self.norm1, p_out)
        return ffn_layer_norm(x, self.linear1, self.linear2, self.norm2, p_ffn)


class _GN8Gelu(torch.autograd.Function):
    """GELU(GroupNorm(h) (+ res)) with 8 channels per group through pmx_gn8_gelu_forward/backward."""

    @staticmethod
    def forward(ctx, h, res, w, b, groups, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        # channels-last bf16 tensors (what MIOpen's NHWC convolutions produce) take the NHWC kernels, anything else NCHW
        cl = h.dtype == torch.bfloat16 and h.is_contiguous(memory_format=torch.channels_last) and not h.is_contiguous()
        fmt = torch.channels_last if cl else torch.contiguous_format
        h = h.contiguous(memory_format=fmt)
        res = res.contiguous(memory_format=fmt) if res is not None else None
        B, Cc = h.shape[0], h.shape[1]
        HW = h.numel() // (B * Cc)
        y = torch.empty_like(h)
        mean = torch.empty(B * groups, dtype=torch.float32, device=h.device)
        rstd = torch.empty(B * groups, dtype=torch.float32, device=h.device)
        wf, bf = w.float().contiguous(), b.float().contiguous()
        st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        rp = res.data_ptr() if res is not None else None
        if cl:
            _lib.check(lib.pmx_gn8cl_gelu_forward(h.data_ptr(), rp, wf.data_ptr(), bf.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                  rstd.data_ptr(), B, groups, HW, float(eps), st), "pmx_gn8cl_gelu_forward")
        else:
            _lib.check(lib.pmx_gn8_gelu_forward(h.data_ptr(), rp, wf.data_ptr(), bf.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), B, groups, HW, float(eps), 0 if h.dtype == torch.float32 else 1, st),
                       "pmx_gn8_gelu_forward")
        ctx.save_for_backward(h, res if res is not None else h.new_empty(0), wf, bf, mean, rstd)
        ctx.has_res, ctx.groups, ctx.HW, ctx.cl, ctx.wdtype = res is not None, groups, HW, cl, w.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        h, res, wf, bf, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last if ctx.cl else torch.contiguous_format)
        B, Cc = h.shape[0], h.shape[1]
        dh = torch.empty_like(h)
        dres = torch.empty_like(h) if ctx.has_res else None
        partial = torch.empty(B, Cc, 2, dtype=torch.float32, device=h.device)
        st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        rp, dp = (res.data_ptr(), dres.data_ptr()) if ctx.has_res else (None, None)
        if ctx.cl:
            _lib.check(lib.pmx_gn8cl_gelu_backward(h.data_ptr(), rp, gy.data_ptr(), wf.data_ptr(), bf.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), dh.data_ptr(), dp, partial.data_ptr(), B, ctx.groups, ctx.HW, st),
                       "pmx_gn8cl_gelu_backward")
        else:
            _lib.check(lib.pmx_gn8_gelu_backward(h.data_ptr(), rp, gy.data_ptr(), wf.data_ptr(), bf.data_ptr(), mean.data_ptr(),
                                                 rstd.data_ptr(), dh.data_ptr(), dp, partial.data_ptr(), B, ctx.groups, ctx.HW,
                                                 0 if h.dtype == torch.float32 else 1, st), "pmx_gn8_gelu_backward")
        g = partial.sum(0).to(ctx.wdtype)
        return dh, dres, g[:, 0], g[:, 1], None, None


def group_norm_gelu(h, res, gn):
    """GELU(gn(h) (+ res)): the fused HIP kernels on the GPU when the group has 8 channels, torch ops otherwise."""
    if (h.is_cuda and h.dim() == 4 and gn.num_channels == 8 * gn.num_groups and h.dtype in (torch.float32, torch.bfloat16)
            and (res is None or res.dtype == h.dtype) and h.shape[2] * h.shape[3] <= 1024):
        return _GN8Gelu.apply(h, res, gn.weight, gn.bias, gn.num_groups, gn.eps)
    z = gn(h)
    return F.gelu(z if res is None else z + res)


class ResidualBlock(nn.Module):
    """conv3x3 - GroupNorm(4) - GELU - conv3x3 - GroupNorm(4) - (+x) - GELU   (pacman_mappo_resnet.py:49-67)"""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1)
        self.gn1 = nn.GroupNorm(4, channels)
        self.act = nn.GELU()
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1)
        self.gn2 = nn.GroupNorm(4, channels)

    def forward(self, x):
        y = group_norm_gelu(self.conv1(x), None, self.gn1)
        return group_norm_gelu(self.conv2(y), x, self.gn2)


class PositionalEncoding2D(nn.Module):
    """Fixed sin/cos table: first half of the channels encodes y, second half x (pacman_mappo_resnet.py:69-95)."""

    def __init__(self, d_model, max_h=50, max_w=50):
        super().__init__()
        half = d_model // 2
        den = torch.exp(torch.arange(0, half, 2) * -(math.log(10000.0) / half))

        def table(n):
            pos = torch.arange(n).unsqueeze(1)
            t = torch.zeros(n, half)
            t[:, 0::2] = torch.sin(pos * den)
            t[:, 1::2] = torch.cos(pos * den)
            return t

        self.register_buffer("y_enc", table(max_h))
        self.register_buffer("x_enc", table(max_w))

    def table(self, H, W):
        """[H, W, d_model]"""
        return torch.cat([self.y_enc[:H, None, :].expand(H, W, -1), self.x_enc[None, :W, :].expand(H, W, -1)], dim=2)

    def forward(self, x):
        _, _, H, W = x.shape
        return x + self.table(H, W).permute(2, 0, 1).unsqueeze(0).to(x.dtype)


class MAPPOAgent(nn.Module):
    """Actor: conv 8->16->32, three residual blocks, Linear(32*H*W -> 512), LayerNorm, GELU, Linear(512 -> 5).
    Critic: conv 8->32 + 2-D positional encoding, 2 post-LN Transformer encoder layers (d=32, 4 heads, ff=128) over
    the H*W tokens, mean pool, Linear 32->512->1 (pacman_mappo_resnet.py:97-170)."""

    def __init__(self, obs_shape, action_dim=5, num_agents=2):
        super().__init__()
        self.obs_shape = tuple(obs_shape)
        C, H, W = self.obs_shape
        ch = 32
        self.actor_backbone = nn.Sequential(
            nn.Conv2d(C, 16, 3, padding=1), nn.GELU(), nn.Conv2d(16, ch, 3, padding=1), nn.GELU(),
            ResidualBlock(ch), ResidualBlock(ch), ResidualBlock(ch), nn.Flatten())
        self.actor_head = nn.Sequential(nn.Linear(ch * H * W, 512), nn.LayerNorm(512), nn.GELU(), nn.Linear(512, action_dim))
        self.d_model = 32
        self.critic_projector = nn.Sequential(nn.Conv2d(C, self.d_model, 3, padding=1))
        self.pos_encoder = PositionalEncoding2D(self.d_model)
        layer = CriticEncoderLayer(d_model=self.d_model, nhead=4, dim_feedforward=128, dropout=0.0, batch_first=False)
        self.critic_transformer = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False)
        self.critic_head = nn.Sequential(nn.Linear(self.d_model, 512), nn.GELU(), nn.Linear(512, 1))
        self.apply(self._init_weights)                                   # :149-158
        nn.init.orthogonal_(self.actor_head[-1].weight, gain=0.01)
        nn.init.orthogonal_(self.critic_head[-1].weight, gain=1.0)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
            if m.bias is not None:
                m.bias.data.fill_(0.0)

    fused_ffn = True        # use the fused feed-forward + LayerNorm kernels of the critic's encoder layers (bf16 on the GPU)
    fused_tower = True      # use the fused actor-tower kernels where they apply (bf16 on the GPU, supported board size)
    batch_major_critic = True   # run the critic channels-last / batch-major under bf16 autocast on the GPU (no transposing copies)
    fused_loss = True       # the PPO objective and its gradient w.r.t. logits / values as one kernel on the GPU (pmx_ppo_loss)
    tower_pack = None       # packed tower parameters for inference, set by a caller that knows the weights are frozen
                            # (VecMAPPOTrainer.rollout); None = pack on every call

    fused_heads = True      # the small ends of the two heads as one kernel each way under bf16 autocast (csrc/pmx_heads.hip)
    two_streams = True      # optimizer step on the GPU: the critic's forward and backward on a side stream, beside the actor's

    def _fused_heads_ok(self, t):
        """The head-tail kernels take bfloat16 activations under bf16 autocast on the GPU and float32 master weights of the
        reference's shapes (they round the linears' operands to bf16 themselves, as autocast does)."""
        ah, ch = self.actor_head, self.critic_head
        return (self.fused_heads and t.is_cuda and t.dtype == torch.bfloat16 and torch.is_autocast_enabled()
                and ah[3].weight.dtype == torch.float32 and ch[0].weight.dtype == torch.float32 and ch[2].weight.dtype == torch.float32
                and tuple(ah[3].weight.shape) == (5, 512) and tuple(ch[0].weight.shape) == (512, 32) and tuple(ch[2].weight.shape) == (1, 512)
                and ah[1].weight.dtype == torch.float32)

    def _use_fused_tower(self, obs):
        if not (self.fused_tower and obs.is_cuda and obs.dim() == 4 and obs.dtype in (torch.bfloat16, torch.uint8)):
            return False
        if self.actor_backbone[0].weight.dtype != torch.float32:
            return False
        from . import actor_tower
        return actor_tower.tower_supported(obs.shape[2], obs.shape[3])

    def logits(self, obs):
        if self._use_fused_tower(obs):
            # one kernel for the whole convolutional tower (csrc/pmx_actor.hip); its output is channels-last, nn.Flatten's
            # order is channel-major
            from . import actor_tower
            if self.tower_pack is not None and not torch.is_grad_enabled():
                feat = actor_tower.tower_forward(obs, self.tower_pack)
            else:
                feat = actor_tower.actor_tower(self.actor_backbone, obs)         # [B, H*W, 32] bf16
            # nn.Flatten's order is (channel, cell), the kernel's is (cell, channel): permute the 5 MB weight of the first
            # head layer instead of the features (160 MB per 16 384 samples, forward and again backward); autograd carries the
            # gradient back through the view
            lin = self.actor_head[0]
            HW = feat.shape[1]
            w = lin.weight.view(lin.out_features, 32, HW).permute(0, 2, 1).reshape(lin.out_features, HW * 32)
            h = F.linear(feat.reshape(feat.shape[0], HW * 32), w, lin.bias)
            if self._fused_heads_ok(h) and h.shape[1] == 512:
                ln, out = self.actor_head[1], self.actor_head[3]
                return _ActorTail.apply(h, ln.weight, ln.bias, out.weight, out.bias, ln.eps)
            return self.actor_head[1:](h)
        if obs.dtype == torch.uint8:            # byte planes are an input format of the fused tower only
            obs = obs.to(torch.bfloat16 if (obs.is_cuda and torch.is_autocast_enabled()) else torch.float32)
        if obs.is_cuda and obs.dtype == torch.bfloat16 and obs.dim() == 4:
            # channels-last end to end: MIOpen's NHWC bf16 implicit-GEMM convolutions and the NHWC GroupNorm kernels then
            # need no layout transposes; nn.Flatten still flattens in logical (C, H, W) order, one copy at the end
            obs = obs.contiguous(memory_format=torch.channels_last)
        return self.actor_head(self.actor_backbone(obs))

    fused_projector = True  # the critic's projector + positional table as one kernel under bf16 autocast (pmx_proj_forward)
    prepack = True          # the encoder layers' parameter packs in one launch at the top of value() instead of one in front of each use

    def _pe_table(self, H, W, dev):
        """The positional table [H*W, 32] float32 on `dev`, built once per board size."""
        key = (H, W, str(dev))
        cache = self.__dict__.setdefault("_pe_cache", {})
        t = cache.get(key)
        if t is None:
            t = cache[key] = self.pos_encoder.table(H, W).reshape(H * W, self.d_model).float().contiguous().to(dev)
        return t

    def _fused_projector_ok(self, merged_obs):
        conv = self.critic_projector[0]
        if not (self.fused_projector and merged_obs.dtype in (torch.uint8, torch.bfloat16, torch.float32) and merged_obs.dim() == 4
                and merged_obs.shape[1] == 8 and conv.weight.dtype == torch.float32 and tuple(conv.weight.shape) == (32, 8, 3, 3)):
            return False
        from . import actor_tower
        return actor_tower.tower_supported(merged_obs.shape[2], merged_obs.shape[3])

    def value(self, merged_obs):
        """merged_obs [B,8,H,W] -> [B] (pacman_mappo_resnet.py:160-170)"""
        layers = self.critic_transformer.layers
        bm = (self.batch_major_critic and merged_obs.is_cuda and torch.is_autocast_enabled() and self.fused_ffn
              and torch.get_autocast_dtype("cuda") == torch.bfloat16
              and layers[0].fused_ok(merged_obs.new_empty(0, dtype=torch.bfloat16), merged_obs.shape[2] * merged_obs.shape[3]))
        if bm:
            # channels-last end to end: the projector's output [B, H, W, d] IS the token tensor [B, S, d] (no transposing copy
            # on the way in, none for its gradient on the way back) and the encoder layers run batch-major
            B, _, H, W = merged_obs.shape
            # every parameter pack of the encoder layers in one launch (they depend on nothing but the weights; six small launches sat in
            # front of the six kernels that read them).  (Making them on a side stream instead crashes hipStreamEndCapture when that
            # stream forks from the critic's side stream -- a fork inside a fork -- on ROCm 7.0: tools/r03_iso.sh.)
            dev = merged_obs.device
            layer_tuple = tuple(layers)
            all_f32 = all(p.dtype == torch.float32 for l in layer_tuple for p in l.parameters()) and len(layer_tuple) <= 4
            fused_proj = self._fused_projector_ok(merged_obs)
            lpacks = encoder_packs(layer_tuple) if (all_f32 and self.prepack) else None
            ppack = None
            if fused_proj:
                conv = self.critic_projector[0]
                x = _Projector.apply(merged_obs, conv.weight, conv.bias, self._pe_table(H, W, dev), ppack)
            else:
                if merged_obs.dtype == torch.uint8:
                    merged_obs = merged_obs.to(torch.bfloat16)
                y = self.critic_projector(merged_obs.contiguous(memory_format=torch.channels_last))
                x = y.permute(0, 2, 3, 1)
                if not x.is_contiguous():
                    x = x.contiguous()
                x = (x + self.pos_encoder.table(H, W).to(x.dtype)).reshape(B, H * W, self.d_model)
            for k, layer in enumerate(layers):
                x = layer.forward_batch_major(x, lpacks[k] if lpacks is not None else None)
            if self.critic_transformer.norm is not None:
                x = self.critic_transformer.norm(x)
            if self._fused_heads_ok(x):
                l1, l2 = self.critic_head[0], self.critic_head[2]
                return _CriticTail.apply(x, l1.weight, l1.bias, l2.weight, l2.bias)
            return self.critic_head(x.mean(dim=1)).squeeze(-1)
        if merged_obs.dtype == torch.uint8:
            merged_obs = merged_obs.to(torch.bfloat16 if (merged_obs.is_cuda and torch.is_autocast_enabled()) else torch.float32)
        x = self.pos_encoder(self.critic_projector(merged_obs))
        x = x.flatten(2).permute(2, 0, 1)                                # [H*W, B, d]
        x = self.critic_transformer(x).mean(dim=0)
        return self.critic_head(x).squeeze(-1)

    def evaluate(self, obs, merged_obs, action):
        """-> value, log_prob, entropy  (:196-205)"""
        # same arithmetic as torch.distributions.Categorical(logits=...): normalise with logsumexp, probs by softmax
        logits = self.logits(obs).float()
        norm = logits - logits.logsumexp(dim=-1, keepdim=True)
        probs = F.softmax(norm, dim=-1)
        logp = norm.gather(1, action.view(-1, 1)).squeeze(1)
        ent = -(norm.clamp(min=torch.finfo(norm.dtype).min) * probs).sum(-1)
        return self.value(merged_obs).float(), logp, ent

    forward = evaluate      # torch.func.functional_call(model, params, (obs, merged, act)) == evaluate with those params

    @torch.no_grad()
    def act(self, obs, generator=None):
        """Sample actions for a batch: -> action [B] int64, log_prob [B] (Categorical(logits).sample, :173-177)."""
        logp_all = F.log_softmax(self.logits(obs).float(), dim=-1)
        a = torch.multinomial(logp_all.exp(), 1, generator=generator).squeeze(1)
        return a, logp_all.gather(1, a.view(-1, 1)).squeeze(1)

    @torch.no_grad()
    def get_deterministic_action(self, obs):                             # :207-211
        return self.logits(obs).argmax(dim=-1)


_RED_ACTION_MAP = (0, 3, 2, 1, 4)   # East <-> West for a red learner (pacman_mappo_resnet.py:232-238)


def canonicalize_action(action, is_red_agent):
    if not is_red_agent:
        return action
    if isinstance(action, torch.Tensor):
        return torch.tensor(_RED_ACTION_MAP, device=action.device, dtype=action.dtype)[action.long()]
    return _RED_ACTION_MAP[int(action)]


class _PPOLossFn(torch.autograd.Function):
    """pmx_ppo_loss: the objective's five scalars from the raw network outputs in one launch; the gradient with respect to the
    logits and the values is computed in the same launch and handed out (scaled) by backward."""

    @staticmethod
    def forward(ctx, logits, values, act, old_logp, adv, ret, clip_eps, ent_coef, vf_coef):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        logits, values = logits.contiguous(), values.contiguous()
        B, BV = logits.shape[0], values.shape[0]
        stats = torch.empty(5, dtype=torch.float32, device=logits.device)
        dlogits, dvalues = torch.empty_like(logits), torch.empty_like(values)

        def scalar(v):
            if isinstance(v, torch.Tensor):
                assert v.is_cuda and v.dtype == torch.float32 and v.numel() == 1
                return v.data_ptr(), 0.0
            return None, float(v)
        cp, ch = scalar(clip_eps)
        ep, eh = scalar(ent_coef)
        st = C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        _lib.check(lib.pmx_ppo_loss(logits.data_ptr(), 1 if logits.dtype == torch.bfloat16 else 0, values.data_ptr(), act.contiguous().data_ptr(),
                                    old_logp.contiguous().data_ptr(), adv.contiguous().data_ptr(), ret.contiguous().data_ptr(), B, BV, cp, ep,
                                    ch, eh, float(vf_coef), stats.data_ptr(), dlogits.data_ptr(), dvalues.data_ptr(), st), "pmx_ppo_loss")
        ctx.save_for_backward(dlogits, dvalues)
        return stats

    @staticmethod
    def backward(ctx, gstats):
        dlogits, dvalues = ctx.saved_tensors
        unit = _UNIT5.get(_dev_key(gstats.device))
        if unit is not None and gstats.data_ptr() == unit.data_ptr():
            return dlogits, dvalues, None, None, None, None, None, None, None      # d(loss) = 1: the kernel's gradients as they are
        g = gstats[4]                                     # only the total loss is an objective; the other entries are reports
        return dlogits * g.to(dlogits.dtype), dvalues * g, None, None, None, None, None, None, None


# Differentiating `loss = stats[4]` costs five small launches before the first real backward kernel (a one for the root, a zero
# 5-vector and the copy of the one into it for the select, two scalings of the kernel's gradients by it) -- 27 us on the critical
# path of a 540 us step.  The learner differentiates the 5-vector itself against this constant [0, 0, 0, 0, 1] instead
# (loss_root), and the backward above recognises it by address.
_UNIT5 = {}


def _dev_key(dev):
    return (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))


def loss_root(loss):
    """(tensor to differentiate, grad_outputs) for a loss returned by ppo_loss: the fused objective's 5-vector with the unit
    gradient where the loss came from the fused kernel, the loss itself otherwise."""
    stats = getattr(loss, "_pmx_stats", None)
    if stats is None:
        return loss, None
    key = _dev_key(stats.device)
    if key not in _UNIT5:
        _UNIT5[key] = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=stats.device)
    return stats, _UNIT5[key]


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra stream per device for the critic half of the optimizer step."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def _fused_loss_ok(model, obs, act, old_logp, adv, ret):
    return (MAPPOAgent.fused_loss and isinstance(model, MAPPOAgent) and obs.is_cuda and act.dtype == torch.int64
            and all(t.dtype == torch.float32 for t in (old_logp, adv, ret)) and act.shape[0] >= 2)


def ppo_loss(model, obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef, vf_coef=VF_COEF):
    """The minibatch objective of pacman_mappo_resnet.py:571-585.  Returns (loss, dict of detached scalars)."""
    if _fused_loss_ok(model, obs, act, old_logp, adv, ret):
        if MAPPOAgent.two_streams:
            # The actor and the critic share nothing until the loss: the critic's forward runs on a side stream, so autograd
            # runs its backward there too (a node's backward uses its forward's stream) and the two halves of the step overlap.
            # At the reference's minibatch of 512 most kernels fill a fraction of the chip and the replayed step was one serial
            # chain of ~80 small launches; in the captured graph the two chains become parallel branches.
            cur = torch.cuda.current_stream(obs.device)
            side = _side_stream(obs.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                vals = model.value(merged).float()
            logits = model.logits(obs)
            cur.wait_stream(side)
            vals.record_stream(cur)
        else:
            logits, vals = model.logits(obs), model.value(merged).float()
        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        stats = _PPOLossFn.apply(logits, vals, act, old_logp, adv, ret, clip_eps, ent_coef, vf_coef)
        d = stats.detach()
        loss = stats[4]
        loss._pmx_stats = stats                           # (loss_root: what the learner differentiates instead of the select)
        return loss, {"pg": d[0], "vl": d[1], "entropy": d[2], "clip_frac": d[3], "loss": d[4]}
    vals, logp, ent = model.evaluate(obs, merged, act)
    if vals.shape[0] != ret.shape[0]:
        # paired minibatch: rows 2k and 2k+1 are the two learners of one env-tick and share ONE merged critic input
        # (merge_obs_for_critic is per env-tick, pacman_mappo_resnet.py:556-557 expands it per agent); the critic ran on
        # the unique inputs only and its value is used for both rows -- the same numbers, half the critic work
        assert vals.shape[0] * 2 == ret.shape[0]
        vals = vals.repeat_interleave(2)
    norm_adv = (adv - adv.mean()) / (adv.std() + 1e-8)                   # unbiased std, per minibatch (:577)
    ratio = (logp - old_logp).exp()
    pg = -torch.min(norm_adv * ratio, norm_adv * torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps)).mean()
    vl = 0.5 * ((vals - ret) ** 2).mean()
    ent_mean = ent.mean()
    loss = pg + vf_coef * vl - ent_coef * ent_mean
    with torch.no_grad():
        clip_frac = ((ratio - 1).abs() > clip_eps).float().mean()
    return loss, {"pg": pg.detach(), "vl": vl.detach(), "entropy": ent_mean.detach(), "clip_frac": clip_frac,
                  "loss": loss.detach()}


import os as _os
if _os.environ.get("PMX_NO_PREPACK"):
    MAPPOAgent.prepack = False
if _os.environ.get("PMX_NO_FUSED_PROJECTOR"):
    MAPPOAgent.fused_projector = False
if _os.environ.get("PMX_NO_TWO_STREAMS"):
    MAPPOAgent.two_streams = False


class FlatBucket:
    """All parameters of a module re-homed as views into one flat fp32 buffer; same for the gradients.
    The data-parallel exchange is then ONE all-reduce of `grad` per optimizer step (2.6 M params = 10.5 MB on
    smallCapture: latency-bound on xGMI, so one message beats many)."""

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        self.data = torch.empty(self.numel, device=dev, dtype=dt)
        self.grad = torch.zeros(self.numel, device=dev, dtype=dt)
        off = 0
        for p in params:
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            off += n
        self.params = params


class PPOLearner:
    """One optimizer step = ppo_loss -> backward into the flat gradient -> (all-reduce mean over ranks) -> clip by global
    norm 0.5 -> Adam(eps 1e-5) -> EMA 0.995 (pacman_mappo_resnet.py:587-595).  Every rank ends each step with bit-identical
    parameters because the clip and the update see the same reduced gradient."""

    def __init__(self, model, lr=LR_START, process_group=None, world_size=1, autocast_dtype=None, force_collectives=False):
        self.model = model
        self.bucket = FlatBucket(model)
        self.ema = self.bucket.data.clone()                              # EMA of the parameters (:365, :593-595)
        self.exp_avg = torch.zeros_like(self.bucket.data)
        self.exp_avg_sq = torch.zeros_like(self.bucket.data)
        self.step_count = 0
        self.lr = lr
        self.betas, self.eps = (0.9, 0.999), 1e-5                        # torch.optim.Adam defaults, eps from :366
        self.pg = process_group
        self.world_size = world_size
        # data parallel: the gradient exchange is issued when there is more than one rank -- or when a caller rehearses the
        # collective path on a one-rank group (bench.py on a one-GPU box)
        self.dp = world_size > 1 or bool(force_collectives)
        self.autocast_dtype = autocast_dtype
        self._w16 = None
        self._sh16 = None
        self._shadow_views = None
        self._graph = self._graphs = None

    def enable_bf16_flat(self):
        """Manual mixed precision instead of autocast for the optimizer step: the network runs on ONE flat bfloat16 copy of
        the float32 master weights (refreshed by one cast kernel per step) whose slices are the functional parameters, so
        that autograd delivers the whole gradient as one flat bfloat16 tensor (the backward of torch.split is a single
        concatenation).  This removes what autocast costs per step on this network: ~47 weight casts, ~46 gradient
        casts back to float32, ~78 accumulate-into-.grad adds and the gradient memset -- about 170 of ~450 launches.  Every
        parameter receives exactly one gradient contribution, and under autocast that contribution was computed in
        bfloat16 as well, so storing it in bfloat16 loses nothing; master weights, Adam moments and EMA stay float32."""
        assert self.bucket.data.is_cuda, "the flat bfloat16 path is a GPU path"
        self._names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        self._sizes = [p.numel() for p in self.bucket.params]
        self._shapes = [tuple(p.shape) for p in self.bucket.params]
        self._w16 = self.bucket.data.to(torch.bfloat16).requires_grad_(True)
        self.autocast_dtype = None

    def _loss_bf16_flat(self, obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef):
        from types import SimpleNamespace
        parts = torch.split(self._w16, self._sizes)
        pd = {n: t.view(sh) for n, t, sh in zip(self._names, parts, self._shapes)}
        fm = SimpleNamespace(evaluate=lambda o, m, a: torch.func.functional_call(self.model, pd, (o, m, a)))
        return ppo_loss(fm, obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef)

    # the second-stage row sums of the gradient reductions folded into the gradient gather (mappo._row_sums_deferred): eight small
    # launches less on the chains of the 512-sample step.  (A first version ran them on a third stream beside the next backward kernel
    # and LOST -- 1 380 against 1 995 steps/s: six fork / join pairs cost the captured graph more than the kernels cost the chain,
    # tools/r03_iso2.sh.)  PMX_NO_DEFER_SUMS=1 restores a kernel per reduction.
    defer_row_sums = not _os.environ.get("PMX_NO_DEFER_SUMS")
    overlap_allreduce = True   # data parallel: reduce the actor's gradient slice while the critic's backward runs
    graph_overlap_allreduce = False   # ... also in the hipGraph-replayed step (one graph per gradient group); see capture()

    def _grad_groups(self):
        """Index ranges [lo, hi) into bucket.params whose gradients are produced -- and, under data parallelism, reduced --
        together, in the order the backward pass is run: the actor's parameters (a contiguous prefix of the bucket that holds
        97 % of its bytes: the 4928 -> 512 head), then the critic's.  One range when nothing is exchanged."""
        n = len(self.bucket.params)
        if not (self.dp and self.overlap_allreduce and self._w16 is None):
            return [(0, n)]
        if getattr(self, "_groups", None) is None:
            names = [k for k, p in self.model.named_parameters() if p.requires_grad]
            n_actor = 0
            while n_actor < n and names[n_actor].startswith("actor_"):
                n_actor += 1
            self._groups = [(0, n_actor), (n_actor, n)] if 0 < n_actor < n else [(0, n)]
            self._offsets = [0]
            for p in self.bucket.params:
                self._offsets.append(self._offsets[-1] + p.numel())
        return self._groups

    def _backward_group(self, loss, lo, hi, retain):
        """Gradients of bucket.params[lo:hi] into their slice of the flat float32 bucket: one gathering copy instead of
        loss.backward() (AccumulateGrad ADDS every parameter's gradient into its zeroed view of the bucket -- ~80 small kernels
        per step on this network, a quarter of the launches of the 512-sample step)."""
        params = self.bucket.params[lo:hi]
        targets = list(params)
        if self._shadow_views is not None:
            for i, v in self._shadow_views.items():
                if lo <= i < hi:
                    targets[i - lo] = v                  # the bfloat16 copy the library op multiplied by
        dev = self.bucket.grad.device
        flat_ok = dev.type == "cuda" and all(t.dtype in (torch.float32, torch.bfloat16) for t in targets)     # (the gather below)
        prev, _DEFER_ROW_SUMS[0] = _DEFER_ROW_SUMS[0], bool(self.defer_row_sums and flat_ok)
        _PENDING_ROWS.clear()                                # (nothing of an earlier, failed call may match this one's addresses)
        root, unit = loss_root(loss)
        try:
            grads = torch.autograd.grad(root, targets, grad_outputs=unit, allow_unused=True, retain_graph=retain)
        finally:
            _DEFER_ROW_SUMS[0] = prev
        flat = [g.reshape(-1) if g is not None else torch.zeros(p.numel(), dtype=self.bucket.grad.dtype, device=dev)
                for g, p in zip(grads, params)]
        first = sum(p.numel() for p in self.bucket.params[:lo])
        if not (dev.type == "cuda" and all(g.dtype in (torch.float32, torch.bfloat16) for g in flat)):
            flush_pending_rows()
        if dev.type == "cuda" and all(g.dtype in (torch.float32, torch.bfloat16) for g in flat):
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            n = len(flat)
            if any(not g.is_contiguous() for g in flat):
                flush_pending_rows()                         # (a copy would read rows that have not been added yet)
            flat = [g.contiguous() for g in flat]
            src = (C.c_void_p * n)(*[g.data_ptr() for g in flat])
            isb = (C.c_uint8 * n)(*[1 if g.dtype == torch.bfloat16 else 0 for g in flat])
            cnt = (C.c_int32 * n)(*[g.numel() for g in flat])
            # gradients that are still partial rows (mappo._row_sums_deferred): the gather adds them
            pend = [pending_rows_of(g.data_ptr()) if g.dtype == torch.float32 else (0, 0) for g in flat]
            rows = (C.c_int32 * n)(*[r for r, _ in pend])
            stride = (C.c_int32 * n)(*[f for _, f in pend])
            offs, o = [], first
            for g in flat:
                offs.append(o); o += g.numel()
            off = (C.c_int64 * n)(*offs)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.pmx_flatten_sum_to_f32(n, src, isb, rows, stride, off, cnt, self.bucket.grad.data_ptr(), st), "pmx_flatten_sum_to_f32")
            _PENDING_ROWS.clear()
        else:
            total = sum(g.numel() for g in flat)
            torch.cat([g.to(self.bucket.grad.dtype) for g in flat], out=self.bucket.grad[first:first + total])

    def _reduce_slice(self, lo, hi):
        """Starts the all-reduce (mean over ranks) of the gradient slice of bucket.params[lo:hi]; returns a closure that makes the
        current stream wait for it and finishes the mean.  RCCL averages in the collective; gloo (the CPU tests) sums, and the
        division follows."""
        import torch.distributed as dist
        self._grad_groups()
        a, b = (self._offsets[lo], self._offsets[hi]) if getattr(self, "_offsets", None) else (0, self.bucket.numel)
        sl = self.bucket.grad[a:b]
        avg = self.bucket.grad.is_cuda
        h = dist.all_reduce(sl, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)

        def finish():
            h.wait()
            if not avg:
                sl.div_(self.world_size)
        return finish

    def _backward_into_bucket(self, loss):
        """The backward pass with the gradient ending up in the flat float32 bucket and, under data parallelism, averaged over the
        ranks: group by group, each group's all-reduce in flight while the next group's backward runs."""
        if self._w16 is not None:
            root, unit = loss_root(loss)
            (g16,) = torch.autograd.grad(root, (self._w16,), grad_outputs=unit)
            self.bucket.grad.copy_(g16)
            if self.dp:
                self._reduce_slice(0, len(self.bucket.params))()
            return
        groups = self._grad_groups()
        pending = []
        for k, (lo, hi) in enumerate(groups):
            self._backward_group(loss, lo, hi, retain=k + 1 < len(groups))
            if self.dp:
                pending.append(self._reduce_slice(lo, hi))
        for fin in pending:
            fin()

    def _refresh_bf16(self):
        if getattr(self, "_bf16_fresh", False):               # the optimizer kernel has just written the (one) copy
            self._bf16_fresh = False
            return
        if self._w16 is not None:
            with torch.no_grad():
                self._w16.copy_(self.bucket.data)
        if self._sh16 is not None:
            with torch.no_grad():
                self._sh16.copy_(self.bucket.data)

    # parameters that only torch library ops read (the two heads' linears, the critic's projector): under bf16 autocast each of them
    # was cast to bfloat16 on use and its gradient cast back -- ~20 small kernels per step.  They are read from a bfloat16 copy of
    # the WHOLE bucket instead (one cast kernel after the optimizer step); the copy's slices take the parameters' places for the
    # forward pass and receive the gradients, which pmx_flatten_to_f32 widens into the bucket.  Same roundings as autocast's.
    SHADOWED = ("actor_head.0.", "actor_head.3.", "critic_projector.0.", "critic_head.0.", "critic_head.2.")
    # (the head-tail kernels and the projector kernel read their layers' float32 masters and round the operands themselves)
    SMALL_HEAD_LAYERS = ("actor_head.3.", "critic_head.0.", "critic_head.2.")

    def _shadowed_prefixes(self):
        keep = self.SHADOWED
        if getattr(self.model, "fused_heads", False):
            keep = tuple(k for k in keep if k not in self.SMALL_HEAD_LAYERS)
        if getattr(self.model, "fused_projector", False):
            keep = tuple(k for k in keep if k != "critic_projector.0.")
        return keep
    shadow_weights = True

    def _shadow_context(self):
        """Context manager under which the module's library-op parameters are their bfloat16 shadows (and a no-op when the
        shadows do not apply: CPU, float32 steps, the flat-bf16 mode)."""
        import contextlib
        if not (self.shadow_weights and self.autocast_dtype == torch.bfloat16 and self._w16 is None and self.bucket.data.is_cuda):
            self._shadow_views = None
            return contextlib.nullcontext()
        if self._sh16 is None:
            self._sh16 = self.bucket.data.to(torch.bfloat16).requires_grad_(True)
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            self._shadow_slots, off = [], 0
            for i, (n, p) in enumerate(zip(names, self.bucket.params)):
                if n.startswith(self._shadowed_prefixes()):
                    self._shadow_slots.append((i, n, off, p.numel(), tuple(p.shape)))
                off += p.numel()
        self._shadow_views, pd = {}, {}
        for i, n, off, k, shape in self._shadow_slots:
            v = self._sh16[off:off + k].view(shape)
            self._shadow_views[i] = v
            pd[n] = v
        from torch.nn.utils import stateless
        return stateless._reparametrize_module(self.model, pd)

    def set_lr(self, lr):
        self.lr = lr

    def _adam_step(self):
        """torch.optim.Adam's single-tensor update (no amsgrad, no weight decay) on the flat buffers."""
        self.step_count += 1
        b1, b2 = self.betas
        g, p = self.bucket.grad, self.bucket.data
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(self.eps)
        p.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    fused_optimizer = True   # clip + Adam + EMA as two launches on the GPU (pmx_clip_adam_ema) instead of ~18 torch kernels

    def _fused_tail(self, scalars_dev=None, reports5=None, report_sums6=None):
        """clip_grad_norm_ -> Adam -> EMA through pmx_clip_adam_ema_tail; returns the gradient norm (a 0-dim device tensor).  With
        scalars_dev the bias-corrected step sizes are read from that device tensor (graph replay), else computed here from
        step_count, which the caller has already advanced.  The same launch refreshes the bfloat16 copy of the parameters (where
        there is exactly one) and, given report_sums6, adds the objective's five scalars and the gradient norm to it."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = self.bucket.data.device
        if getattr(self, "_opt_scratch", None) is None:
            self._opt_scratch = torch.empty(_lib.OPT_PARTIALS, dtype=torch.float64, device=dev)
            self._opt_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        b1, b2 = self.betas
        if scalars_dev is None:
            bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
            sp, a, b = None, self.lr / bc1, 1.0 / math.sqrt(bc2)
        else:
            sp, a, b = scalars_dev.data_ptr(), 0.0, 0.0
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        copies = [t for t in (self._w16, self._sh16) if t is not None]
        p16 = copies[0] if len(copies) == 1 and copies[0].is_contiguous() and copies[0].numel() == self.bucket.numel else None
        _lib.check(lib.pmx_clip_adam_ema_tail(self.bucket.grad.data_ptr(), self.bucket.data.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.ema.data_ptr(), self.bucket.numel, self._opt_scratch.data_ptr(),
                                              sp, a, b, b1, b2, self.eps, MAX_GRAD_NORM, EMA_DECAY, self._opt_norm.data_ptr(),
                                              p16.data_ptr() if p16 is not None else None,
                                              reports5.data_ptr() if reports5 is not None else None,
                                              report_sums6.data_ptr() if report_sums6 is not None else None, st), "pmx_clip_adam_ema_tail")
        self._bf16_fresh = p16 is not None
        return self._opt_norm[0]

    def _use_fused_tail(self):
        return (self.fused_optimizer and self.bucket.data.is_cuda and self.bucket.data.dtype == torch.float32
                and self.bucket.grad.is_contiguous() and self.bucket.data.is_contiguous())

    def update_minibatch(self, obs, merged, act, old_logp, adv, ret, clip_eps=CLIP_EPS, ent_coef=ENT_COEF_START):
        dev_type = self.bucket.data.device.type
        if self._w16 is not None:
            loss, stats = self._loss_bf16_flat(obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef)
        elif self.autocast_dtype is not None:
            with self._shadow_context(), torch.autocast(device_type=dev_type, dtype=self.autocast_dtype):
                loss, stats = ppo_loss(self.model, obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef)
        else:
            self._shadow_views = None
            loss, stats = ppo_loss(self.model, obs, merged, act, old_logp, adv, ret, clip_eps, ent_coef)
        self._backward_into_bucket(loss)                  # (data parallel: includes the gradient all-reduce)
        # clip_grad_norm_(parameters, 0.5): 2-norm of the per-tensor 2-norms (the reference's summation order; a single
        # fp32 reduction over the 2.6 M-element flat buffer is measurably less accurate on the CPU), then scale by
        # max_norm / (norm + 1e-6) if that is < 1
        if self._use_fused_tail():
            self.step_count += 1
            gn = self._fused_tail().clone()
        else:
            gn = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([p.grad for p in self.bucket.params])))
            self.bucket.grad.mul_(torch.clamp(MAX_GRAD_NORM / (gn + 1e-6), max=1.0))
            self._adam_step()
            self.ema.mul_(EMA_DECAY).add_(self.bucket.data, alpha=1 - EMA_DECAY)
        self._refresh_bf16()
        stats["grad_norm"] = gn.detach()
        return stats

    # ---- hipGraph capture of the whole optimizer step (launch-bound at the reference's minibatch of 512) ----------
    @staticmethod
    def graph_replay_safe():
        """hipGraph replay is only trusted with ROCclr's AQL packet capture switched off, and only if the runtime can have
        read the switch: set before HIP initialised (see the package __init__, which records that at import)."""
        import os
        import sys
        pkg = sys.modules.get(__name__.rsplit(".", 1)[0])
        return os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and bool(getattr(pkg, "GRAPH_REPLAY_OK", False))

    def capture(self, batch, obs_shape, in_dtype, clip_eps=CLIP_EPS, ent_coef=ENT_COEF_START, merged_batch=None):
        """Record zero_grad -> forward -> backward -> (all-reduce) -> clip -> Adam -> EMA for a fixed minibatch shape into
        a HIP graph.  Scalars that change between updates (lr, clip, entropy coefficient, Adam bias corrections) live in
        device tensors that the graph reads, so one capture serves the whole schedule."""
        if not self.graph_replay_safe():
            raise RuntimeError("hipGraph replay of the optimizer step needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment "
                               "before HIP initialises (ROCm 7 graph packet-capture bug, DESIGN.md section 5)")
        dev = self.bucket.data.device
        self._g_in = dict(obs=torch.zeros((batch,) + tuple(obs_shape), dtype=in_dtype, device=dev),
                          merged=torch.zeros((merged_batch or batch,) + tuple(obs_shape), dtype=in_dtype, device=dev),
                          act=torch.zeros(batch, dtype=torch.int64, device=dev),
                          logp=torch.zeros(batch, dtype=torch.float32, device=dev),
                          adv=torch.randn(batch, dtype=torch.float32, device=dev),
                          ret=torch.zeros(batch, dtype=torch.float32, device=dev))
        # lr/bc1, 1/sqrt(bc2), clip_eps, ent_coef as device scalars
        self._g_sc = torch.zeros(4, dtype=torch.float32, device=dev)
        self._g_stats = None
        self._g_acc = torch.zeros(6, dtype=torch.float32, device=dev)       # sums of (pg, vl, entropy, clip_frac, loss, grad_norm)

        def seg_loss():
            i = self._g_in
            if self._w16 is not None:
                loss, stats = self._loss_bf16_flat(i["obs"], i["merged"], i["act"], i["logp"], i["adv"], i["ret"],
                                                   self._g_sc[2], self._g_sc[3])
            elif self.autocast_dtype is not None:
                with self._shadow_context(), torch.autocast(device_type=dev.type, dtype=self.autocast_dtype):
                    loss, stats = ppo_loss(self.model, i["obs"], i["merged"], i["act"], i["logp"], i["adv"], i["ret"],
                                           self._g_sc[2], self._g_sc[3])
            else:
                self._shadow_views = None
                loss, stats = ppo_loss(self.model, i["obs"], i["merged"], i["act"], i["logp"], i["adv"], i["ret"],
                                       self._g_sc[2], self._g_sc[3])
            return loss, stats

        REPORTS = ("pg", "vl", "entropy", "clip_frac", "loss")

        def seg_tail(stats):
            # the fused objective's 5-vector (ppo_loss leaves it on the loss): with it the optimizer kernel keeps the running sums
            s5 = getattr(state.get("loss"), "_pmx_stats", None)
            in_kernel = self._use_fused_tail() and s5 is not None and tuple(stats.keys()) == REPORTS and s5.dtype == torch.float32
            if self._use_fused_tail():
                gn = self._fused_tail(self._g_sc, *((s5.detach(), self._g_acc) if in_kernel else ()))
            else:
                self._g_norms = torch.stack(torch._foreach_norm([p.grad for p in self.bucket.params]))   # kept: per-tensor norms
                gn = torch.linalg.vector_norm(self._g_norms)
                self.bucket.grad.mul_(torch.clamp(MAX_GRAD_NORM / (gn + 1e-6), max=1.0))
                b1, b2 = self.betas
                g, p = self.bucket.grad, self.bucket.data
                self.exp_avg.lerp_(g, 1 - b1)
                self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.exp_avg_sq.sqrt() * self._g_sc[1]).add_(self.eps)
                p.sub_(self.exp_avg / denom * self._g_sc[0])
                self.ema.mul_(EMA_DECAY).add_(p, alpha=1 - EMA_DECAY)
            self._refresh_bf16()
            stats["grad_norm"] = gn
            # running sums of the step's reports, inside the graph: a caller that averages them over an update reads ONE tensor at
            # the end instead of launching an add per report and step
            self._g_acc_keys = tuple(stats.keys())
            if not in_kernel:
                self._g_acc.add_(torch.stack([stats[k].float() for k in self._g_acc_keys]))
            return stats

        # Data parallel: the collectives stay OUTSIDE the graphs (eager RCCL calls, exactly the ones the eager step issues), so
        # the step is recorded as one graph per gradient group plus one for the optimizer tail (with graph_overlap_allreduce):
        #   graph 0: forward, loss, backward of group 0 (the actor), its gradients into the bucket   | all-reduce of slice 0 starts
        #   graph 1: backward of group 1 (the critic), its gradients into the bucket                 | ... overlaps this graph
        #   graph 2: clip, Adam, EMA, weight copies, reports                                          | after both all-reduces
        # The autograd graph built while graph 0 is recorded is walked again while graph 1 is recorded; all graphs share one
        # memory pool, so what graph 0 saved for the backward pass stays where graph 1's kernels read it.
        # Replayed steps are the launch-bound ones (small minibatches): there the actor's and the critic's backward run side by side on
        # two streams inside one graph, which buys more than reducing the actor's slice early would (1 800 against 1 300 steps/s at 512
        # samples on one GPU), so the replayed data-parallel step is [forward + whole backward] -> ONE all-reduce -> [optimizer tail].
        # graph_overlap_allreduce = True restores one graph per gradient group (the eager step always reduces slice by slice).
        groups = self._grad_groups() if (self._w16 is None and self.graph_overlap_allreduce) else [(0, len(self.bucket.params))]
        segmented = self.dp
        state = {}

        def seg_first():
            state["loss"], state["stats"] = seg_loss()
            if self._w16 is not None:
                root, unit = loss_root(state["loss"])
                (g16,) = torch.autograd.grad(root, (self._w16,), grad_outputs=unit)
                self.bucket.grad.copy_(g16)
            else:
                self._backward_group(state["loss"], *groups[0], retain=len(groups) > 1)

        def seg_group(k):
            self._backward_group(state["loss"], *groups[k], retain=k + 1 < len(groups))

        def run_eager():
            seg_first()
            pend = [self._reduce_slice(*groups[0])] if self.dp else []
            for k in range(1, len(groups)):
                seg_group(k)
                if self.dp:
                    pend.append(self._reduce_slice(*groups[k]))
            for fin in pend:
                fin()
            return seg_tail(state["stats"])

        # warm up on a side stream (allocator, MIOpen solver search), restoring the optimizer state afterwards
        saved = [t.clone() for t in (self.bucket.data, self.exp_avg, self.exp_avg_sq, self.ema)]
        self._set_graph_scalars(clip_eps, ent_coef, step=1)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                run_eager()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        state.clear()
        # Capture in THREAD-LOCAL error mode: in the default (global) mode a capture in progress forbids "unsafe" runtime calls from
        # every thread of the process -- and ProcessGroupNCCL's watchdog thread polls the events of recent collectives with
        # hipEventQuery, which then fails with hipErrorStreamCaptureUnsupported and takes the process down (seen when a capture began
        # right behind the warm-up steps' all-reduces).  This thread makes no such call while it captures; other threads' launches on
        # the capturing streams (autograd's workers) are recorded either way.  The pause lets the watchdog retire what has completed.
        if self.dp:
            import time as _time
            _time.sleep(0.25)
        tl = dict(capture_error_mode="thread_local")
        if not segmented:
            self._graph = torch.cuda.CUDAGraph()
            self._graphs = None
            with torch.cuda.graph(self._graph, **tl):
                seg_first()
                for k in range(1, len(groups)):
                    seg_group(k)
                self._g_stats = seg_tail(state["stats"])
        else:
            self._graphs, pool = [], None
            for k in range(len(groups) + 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, **tl):
                    if k == 0:
                        seg_first()
                    elif k < len(groups):
                        seg_group(k)
                    else:
                        self._g_stats = seg_tail(state["stats"])
                pool = g.pool()
                self._graphs.append(g)
            self._graph = None
            self._g_groups = list(groups)
        state.clear()
        for t, v in zip((self.bucket.data, self.exp_avg, self.exp_avg_sq, self.ema), saved):
            t.copy_(v)
        self._refresh_bf16()
        self._g_batch = batch

    def _replay(self):
        """Replays the captured step; data parallel: graph per gradient group with that group's all-reduce started behind it (on
        RCCL's own stream, so it runs beside the next graph), then the optimizer tail."""
        if self._graphs is None:
            self._graph.replay()
            return
        pend = []
        for k, (lo, hi) in enumerate(self._g_groups):
            self._graphs[k].replay()
            pend.append(self._reduce_slice(lo, hi))
        for fin in pend:
            fin()
        self._graphs[-1].replay()

    def _graph_scalar_values(self, clip_eps, ent_coef, step):
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        return tuple(float(v) for v in (self.lr / bc1, 1.0 / math.sqrt(bc2), clip_eps, ent_coef))

    def _set_graph_scalars(self, clip_eps, ent_coef, step):
        vals = self._graph_scalar_values(clip_eps, ent_coef, step)
        if self._g_sc.is_cuda:
            # the four values travel as kernel arguments of ONE launch: stream-ordered, no host staging buffer that could be
            # recycled while a copy is still in flight
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            arr = (C.c_float * 4)(*[float(v) for v in vals])
            st = C.c_void_p(torch.cuda.current_stream(self._g_sc.device).cuda_stream)
            _lib.check(lib.pmx_set_floats(self._g_sc.data_ptr(), arr, 4, st), "pmx_set_floats")
            return
        for k, v in enumerate(vals):
            self._g_sc[k].fill_(float(v))

    def update_minibatch_graph(self, obs, merged, act, old_logp, adv, ret, clip_eps=CLIP_EPS, ent_coef=ENT_COEF_START):
        """Same step as update_minibatch, replayed from the captured graph (inputs are copied into its static buffers)."""
        i = self._g_in
        i["obs"].copy_(obs); i["merged"].copy_(merged); i["act"].copy_(act)
        i["logp"].copy_(old_logp); i["adv"].copy_(adv); i["ret"].copy_(ret)
        self.step_count += 1
        self._set_graph_scalars(clip_eps, ent_coef, self.step_count)
        self._replay()
        return self._g_stats

    def update_minibatch_graph_gather(self, sources, index, rows_per_index, clip_eps=CLIP_EPS, ent_coef=ENT_COEF_START):
        """The replayed step with its minibatch assembled by ONE gather launch (pmx_gather_rows) straight into the graph's
        static inputs: sources = the flat rollout tensors {obs, merged, act, logp, adv, ret} (rows = samples; merged rows =
        env-ticks), index = int64 device indices, rows_per_index = {name: m} (2 for the per-learner tensors of a paired
        minibatch whose index names env-tick pairs).  Returns the graph's stats tensors (valid until the next replay)."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        names = ("obs", "merged", "act", "logp", "adv", "ret")
        n = len(names)
        VP, I32, I64 = C.c_void_p * n, C.c_int32 * n, C.c_int64 * n
        src, dst, idx, rb, m, nr = VP(), VP(), VP(), I32(), I32(), I64()
        for k, name in enumerate(names):
            s_t, d_t = sources[name], self._g_in[name]
            assert s_t.is_contiguous() and d_t.is_contiguous() and s_t.dtype == d_t.dtype and s_t.shape[1:] == d_t.shape[1:], name
            row = d_t[0].numel() * d_t.element_size() if d_t.dim() > 1 else d_t.element_size()
            src[k], dst[k], idx[k] = s_t.data_ptr(), d_t.data_ptr(), index.data_ptr()
            rb[k], m[k], nr[k] = row, rows_per_index[name], d_t.shape[0]
            assert index.numel() * rows_per_index[name] == d_t.shape[0], (name, index.numel(), d_t.shape)
        assert index.dtype == torch.int64 and index.is_contiguous()
        st = C.c_void_p(torch.cuda.current_stream(index.device).cuda_stream)
        self.step_count += 1
        vals = self._graph_scalar_values(clip_eps, ent_coef, self.step_count)        # (they travel with the gather: one launch)
        arr = (C.c_float * 4)(*vals)
        _lib.check(lib.pmx_gather_rows_set_floats(n, src, dst, idx, rb, m, nr, self._g_sc.data_ptr(), arr, 4, st), "pmx_gather_rows_set_floats")
        self._replay()
        return self._g_stats

    def ema_state_dict(self):
        """state_dict of the EMA weights with the reference's parameter names (what :647-651 saves)."""
        sd = {k: v.clone() for k, v in self.model.state_dict().items()}
        off = 0
        names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        for n, p in zip(names, self.bucket.params):
            k = p.numel()
            sd[n] = self.ema[off:off + k].view_as(p).clone()
            off += k
        return sd


def schedule(update, total_updates):
    """Linear lr / entropy-coefficient anneal and the clip switch (pacman_mappo_resnet.py:386-391)."""
    progress = update / total_updates
    lr = LR_START - (LR_START - LR_END) * progress
    ent = ENT_COEF_START - (ENT_COEF_START - ENT_COEF_END) * progress
    clip = 0.1 if update > 800 else CLIP_EPS
    return lr, ent, clip
