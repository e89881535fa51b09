"""ctypes binding of the C ABI in include/pmx.h (libpmx_hip.so).

There is no CPU fallback: if the HIP library is missing or does not load, importing the product fails loudly."""
import ctypes as C
import os

from . import build as _build

PMX_MAX_DIM = 32
OBS_F32, OBS_BF16, OBS_U8 = 0, 1, 2
ACTION_RANDOM_LEGAL = -2
ACTION_BASELINE_OFFENSE, ACTION_BASELINE_DEFENSE = -3, -4
COLSUM_BLOCKS = 512
LN32_PARTIAL_ROWS = 2048
ACTOR_PACK_BYTES = 297984
ACTOR_GRAD_FLOATS = 74496
GRAD_PARTIAL_ROWS = 512
OPT_PARTIALS = 1024                # PMX_OPT_PARTIALS: doubles of scratch pmx_clip_adam_ema needs
FFN_PACK_BYTES = 33664
FFN_GRAD_FLOATS = 8416
TOK96_PACK_BYTES, TOK96_GRAD_FLOATS = 12928, 3232
TOK32_PACK_BYTES, TOK32_GRAD_FLOATS = 4480, 1120
HEADS_PARTIAL_ROWS, ACTOR_TAIL_GRAD_FLOATS, CRITIC_TAIL_GRAD_FLOATS = 128, 3592, 17416
PROJ_PACK_BYTES, PROJ_PARTIAL_ROWS, PROJ_GRAD_ROW_FLOATS = 6272, 256, 2592


class PmxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("wall_rows", C.POINTER(C.c_uint32)), ("food_rows", C.POINTER(C.c_uint32)),
                ("cap_rows", C.POINTER(C.c_uint32)), ("starts", C.POINTER(C.c_int8)),
                ("n_envs", C.c_int32), ("length", C.c_int32), ("legal_reward", C.c_int32),
                ("defence_reward", C.c_int32), ("auto_reset", C.c_int32), ("obs_dtype", C.c_int32),
                ("obs_agents", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint32), ("n_layouts", C.c_int32),
                ("layout_index", C.POINTER(C.c_int32)), ("enable_bots", C.c_int32), ("redraw_layouts", C.c_int32)]


class StepOut(C.Structure):
    _fields_ = [("obs_dev", C.c_void_p), ("reward_dev", C.c_void_p), ("done_dev", C.c_void_p),
                ("legal_dev", C.c_void_p), ("score_change_dev", C.c_void_p), ("score_dev", C.c_void_p),
                ("agent_dev", C.c_void_p)]


class State(C.Structure):
    _fields_ = [("pos", (C.c_int8 * 2) * 4), ("dir", C.c_int8 * 4), ("pac", C.c_uint8 * 4), ("scared", C.c_uint8 * 4),
                ("carry", C.c_uint16 * 4), ("ret", C.c_uint16 * 4), ("food", C.c_uint32 * PMX_MAX_DIM),
                ("caps", C.c_uint32 * PMX_MAX_DIM), ("score", C.c_int32), ("steps", C.c_int32), ("ticks", C.c_uint32)]


class ActorParams(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 8), ("conv_b", C.c_void_p * 8), ("gn_w", C.c_void_p * 6), ("gn_b", C.c_void_p * 6)]


class EncoderLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "norm1_w", "norm1_b", "lin1_w", "lin1_b",
                                          "lin2_w", "lin2_b", "norm2_w", "norm2_b", "pack_in", "pack_out", "pack_ffn")]


# every symbol include/pmx.h declares: (name, restype, argtypes)
_VP, _I32 = C.c_void_p, C.c_int32
PROTOTYPES = [
    ("pmx_version", C.c_int, []),
    ("pmx_last_error", C.c_char_p, []),
    ("pmx_create", C.c_int, [C.POINTER(Config), C.POINTER(_VP)]),
    ("pmx_destroy", C.c_int, [_VP]),
    ("pmx_obs_shape", C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    ("pmx_reset", C.c_int, [_VP, _VP, C.POINTER(StepOut), _VP]),
    ("pmx_step", C.c_int, [_VP, _VP, C.POINTER(StepOut), _VP]),
    ("pmx_step_agent", C.c_int, [_VP, C.c_int, _VP, C.POINTER(StepOut), _VP]),
    ("pmx_successor", C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    ("pmx_observe", C.c_int, [_VP, _VP, _VP, _VP]),
    ("pmx_emit_team_obs", C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    ("pmx_get_layout_index", C.c_int, [_VP, C.POINTER(_I32), _VP]),
    ("pmx_get_state", C.c_int, [_VP, _I32, _I32, C.POINTER(State), _VP]),
    ("pmx_set_state", C.c_int, [_VP, _I32, _I32, C.POINTER(State), _VP]),
    ("pmx_maze_distances", C.c_int, [_VP, _VP, _VP, C.POINTER(_I32), _VP]),
    ("pmx_maze_distances_layout", C.c_int, [_VP, _I32, _VP, _VP, C.POINTER(_I32), _VP]),
    ("pmx_profile_begin", C.c_int, [_VP, _I32]),
    ("pmx_set_tuning", C.c_int, [_VP, C.c_char_p, _I32]),
    ("pmx_profile_end", C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I32), C.POINTER(C.c_double), C.POINTER(_I32)]),
    ("pmx_gae", C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, C.c_double, C.c_double, _VP, _VP, _VP]),
    ("pmx_ln32_forward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_float, _I32, _VP]),
    ("pmx_ln32_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _VP]),
    ("pmx_gn8_gelu_forward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, C.c_float, _I32, _VP]),
    ("pmx_gn8_gelu_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _I32, _VP]),
    ("pmx_gn8cl_gelu_forward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, C.c_float, _VP]),
    ("pmx_gn8cl_gelu_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    ("pmx_attn8_forward", C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP]),
    ("pmx_attn8_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _VP]),
    ("pmx_ppo_loss", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, C.c_float, C.c_float, C.c_float, _VP, _VP, _VP, _VP]),
    ("pmx_clip_adam_ema", C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, _VP, _VP]),
    ("pmx_clip_adam_ema_tail", C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_flatten_to_f32", C.c_int, [_I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_flatten_sum_to_f32", C.c_int, [_I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_gather_rows", C.c_int, [_I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_gather_rows_set_floats", C.c_int, [_I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    ("pmx_set_floats", C.c_int, [_VP, _VP, _I32, _VP]),
    ("pmx_attn8_forward_layout", C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    ("pmx_attn8_backward_layout", C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    ("pmx_colsum_bf16", C.c_int, [_VP, C.c_int64, _I32, _VP, _VP]),
    ("pmx_canonicalize_obs", C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    ("pmx_merge_obs", C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    ("pmx_actor_supported", C.c_int, [_I32, _I32]),
    ("pmx_actor_sizes", C.c_int, [_I32, _I32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("pmx_actor_pack", C.c_int, [C.POINTER(ActorParams), _VP, _VP]),
    ("pmx_actor_forward", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    ("pmx_actor_backward", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    ("pmx_actor_unpack_grads", C.c_int, [_VP, C.POINTER(ActorParams), _VP]),
    ("pmx_ffn_pack", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_ffn_forward", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_float, _VP]),
    ("pmx_ffn_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_float, _VP]),
    ("pmx_tok96_pack", C.c_int, [_VP, _VP, _VP, _VP]),
    ("pmx_tok96_forward", C.c_int, [_VP, _VP, _VP, C.c_int64, _VP]),
    ("pmx_tok96_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("pmx_tok96_backward_res", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("pmx_tok32ln_pack", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    ("pmx_tok32ln_forward", C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_float, _VP]),
    ("pmx_tok32ln_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_float, _VP]),
    ("pmx_actor_tail_forward", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_float, _VP]),
    ("pmx_actor_tail_backward", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("pmx_critic_tail_forward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _VP]),
    ("pmx_critic_tail_backward", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, _VP]),
    ("pmx_encoder_pack", C.c_int, [_I32, C.POINTER(EncoderLayerParams), _VP]),
    ("pmx_proj_pack", C.c_int, [_VP, _VP, _VP, _VP]),
    ("pmx_proj_forward", C.c_int, [_VP, _I32, _VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    ("pmx_proj_backward", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    ("pmx_defer_row_sums", C.c_int, [_I32]),
    ("pmx_last_partial_rows", C.c_int, []),
    ("pmx_sum_partial_rows", C.c_int, [_VP, _I32, _I32, _VP]),
]
# test / bench hooks that are not part of the public header
EXTRA = [
    ("pmx_gae_mode", C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, C.c_double, C.c_double, _VP, _VP, C.c_int, _VP]),
]

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load libpmx_hip.so (building it first if the sources are newer and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if _build.stale():
        # the sources changed (or nothing was built yet): rebuild, and let a compile error surface -- a library built from
        # other sources is never used in its place
        try:
            _build.build()
        except Exception as e:
            raise PmxError(f"libpmx_hip.so is missing or out of date and could not be built ({e}); run "
                           f"`python -c 'import __graft_entry__ as g; g.build()'` on a machine with hipcc") from e
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise PmxError(f"cannot load {path}: {e} (the product has no CPU fallback)") from e
    lib.pmx_source_hash.restype = C.c_char_p
    built_from = lib.pmx_source_hash().decode()
    if built_from != _build.source_hash():
        raise PmxError(f"{path} was built from other sources (hash {built_from}, tree {_build.source_hash()}): rebuild it")
    for name, res, args in PROTOTYPES + EXTRA:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pmx_last_error()
        raise PmxError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
