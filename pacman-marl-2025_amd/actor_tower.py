"""The actor's convolutional tower (MAPPOAgent.actor_backbone, pacman_mappo_resnet.py:104-113 with ResidualBlock :49-67)
through the fused HIP kernels pmx_actor_forward / pmx_actor_backward (csrc/pmx_actor.hip): one kernel for the eight
convolutions, GroupNorms and GELUs of the forward pass and one for the whole backward pass, bf16 on the matrix cores.

The module's parameters stay what they are (float32 nn.Parameters with the reference's names); each call packs them into
the kernels' operand layout (one small kernel) and, in training, hands autograd float32 gradients in the parameters'
own shapes.  `tower_supported(H, W)` says whether a board size has a kernel; callers keep the library convolutions
(MIOpen) otherwise and on the CPU."""
import ctypes as C

import torch

from . import _lib

_OBS_CODE = {torch.float32: _lib.OBS_F32, torch.bfloat16: _lib.OBS_BF16, torch.uint8: _lib.OBS_U8}


def tower_supported(H, W):
    return bool(_lib.load().pmx_actor_supported(int(H), int(W)))


def _tower_params(backbone):
    """The 28 parameter tensors in the kernels' order: conv w/b of the 8 layers, then gn w/b of gn1, gn2 of each block."""
    convs = [backbone[0], backbone[2]]
    gns = []
    for blk in (backbone[4], backbone[5], backbone[6]):
        convs += [blk.conv1, blk.conv2]
        gns += [blk.gn1, blk.gn2]
    return [c.weight for c in convs] + [c.bias for c in convs] + [g.weight for g in gns] + [g.bias for g in gns]


def _param_struct(tensors):
    ps = _lib.ActorParams()
    for i in range(8):
        ps.conv_w[i] = tensors[i].data_ptr()
        ps.conv_b[i] = tensors[8 + i].data_ptr()
    for i in range(6):
        ps.gn_w[i] = tensors[16 + i].data_ptr()
        ps.gn_b[i] = tensors[22 + i].data_ptr()
    return ps


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_params(tensors):
    """-> uint8 tensor [PMX_ACTOR_PACK_BYTES] with the MFMA operand fragments of the given parameter tensors."""
    lib = _lib.load()
    ts = [t.detach().float().contiguous() for t in tensors]
    dev = ts[0].device
    pack = torch.empty(_lib.ACTOR_PACK_BYTES, dtype=torch.uint8, device=dev)
    ps = _param_struct(ts)
    _lib.check(lib.pmx_actor_pack(C.byref(ps), pack.data_ptr(), _stream(dev)), "pmx_actor_pack")
    return pack


def _sizes(H, W, B):
    """-> (save bytes, backward scratch bytes, inference scratch bytes) for B samples"""
    lib = _lib.load()
    sv, sc, si = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.pmx_actor_sizes(H, W, B, C.byref(sv), C.byref(sc), C.byref(si)), "pmx_actor_sizes")
    return sv.value, sc.value, si.value


_infer_scratch = {}


def _infer_scratch_buffer(dev, nbytes):
    """The inference kernels' skip-input slots: one buffer per device whose size depends on the board only (2 048 slots), so
    it is allocated once per board size and never replaced by a larger batch.  Kernels on one stream use it one after the
    other; callers that run inference on several streams of one device at once must pass their own `scratch`."""
    key = (dev, int(nbytes))
    buf = _infer_scratch.get(key)
    if buf is None:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _infer_scratch[key] = buf
    return buf


def tower_forward(obs, pack, save=None, scratch=None):
    """obs [B,8,H,W] (uint8 / bfloat16 / float32, contiguous) -> features [B, H*W, 32] bfloat16 (channels-last)."""
    lib = _lib.load()
    B, _, H, W = obs.shape
    obs = obs.contiguous()
    feat = torch.empty((B, H * W, 32), dtype=torch.bfloat16, device=obs.device)
    if save is None and scratch is None:
        scratch = _infer_scratch_buffer(obs.device, _sizes(H, W, B)[2])
    _lib.check(lib.pmx_actor_forward(obs.data_ptr(), _OBS_CODE[obs.dtype], pack.data_ptr(), feat.data_ptr(),
                                     save.data_ptr() if save is not None else None,
                                     scratch.data_ptr() if scratch is not None else None, B, H, W, _stream(obs.device)),
               "pmx_actor_forward")
    return feat


class _ActorTower(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, *params):
        B, _, H, W = obs.shape
        obs = obs.contiguous()
        pack = pack_params(params)
        save = torch.empty(_sizes(H, W, B)[0], dtype=torch.uint8, device=obs.device)
        feat = tower_forward(obs, pack, save)
        ctx.save_for_backward(obs, pack, save)
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.dtypes = [p.dtype for p in params]
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = _lib.load()
        obs, pack, save = ctx.saved_tensors
        B, _, H, W = obs.shape
        dev = obs.device
        dfeat = dfeat.to(torch.bfloat16).contiguous()
        # backward's scratch belongs to THIS call (a stream-ordered allocation from the caching allocator; under hipGraph capture
        # it comes from the graph's private pool and lives as long as the graph).  A module-global grow-only buffer could be
        # replaced -- and its old block handed to another tensor -- by a later, larger call while a captured graph still
        # replays a backward that writes through the old pointer.
        scratch = torch.empty(_sizes(H, W, B)[1], dtype=torch.uint8, device=dev)
        grad = torch.empty(_lib.ACTOR_GRAD_FLOATS, dtype=torch.float32, device=dev)
        _lib.check(lib.pmx_actor_backward(obs.data_ptr(), _OBS_CODE[obs.dtype], pack.data_ptr(), save.data_ptr(), dfeat.data_ptr(),
                                          scratch.data_ptr(), grad.data_ptr(), B, H, W, _stream(dev)), "pmx_actor_backward")
        outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        ps = _param_struct(outs)
        _lib.check(lib.pmx_actor_unpack_grads(grad.data_ptr(), C.byref(ps), _stream(dev)), "pmx_actor_unpack_grads")
        return (None,) + tuple(o.to(dt) for o, dt in zip(outs, ctx.dtypes))


def actor_tower(backbone, obs):
    """actor_backbone(obs) up to (not including) nn.Flatten, as [B, H*W, 32] bfloat16; differentiable in the parameters."""
    params = _tower_params(backbone)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _ActorTower.apply(obs, *params)
    return tower_forward(obs, pack_params(params))
