"""PmxVecEnv: N independent Capture-the-Flag games advanced in lock-step by the HIP kernels.

Tensor-level mirror of gymPacMan.gymPacMan_parallel_env (gymPacMan.py:15-270): same constructor meaning, same
step()/reset() results, but every result carries a leading env dimension and lives on the GPU.  PyTorch is used
only for device memory and the current stream; all compute is in libpmx_hip.so through the C ABI.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .layout import Layout, get_layout

_DTYPES = {"float32": (_lib.OBS_F32, torch.float32), "bfloat16": (_lib.OBS_BF16, torch.bfloat16),
           "uint8": (_lib.OBS_U8, torch.uint8)}
# reference order of a legal-action list: N, S, E, W, Stop as ints 0,2,1,3,4 (game.py:295-299, gymPacMan.py:78-89)
LEGAL_LIST_ORDER = (0, 2, 1, 3, 4)


def legal_list(mask):
    return [a for a in LEGAL_LIST_ORDER if (int(mask) >> a) & 1]


class PmxVecEnv:
    def __init__(self, layout, n_envs, length=299, reward_forLegalAction=True, defenceReward=True, auto_reset=True,
                 obs_dtype="float32", obs_agents=(0, 1, 2, 3), device="cuda:0", seed=0, layout_index=None, bots=False,
                 redraw_layouts=False):
        if not torch.cuda.is_available():
            raise _lib.PmxError("PmxVecEnv needs a GPU: the product has no CPU path")
        self.lib = _lib.load()
        # `layout` may be one layout (name / Layout) or a list of equally sized Layouts with `layout_index` [n_envs]
        layouts = None
        if isinstance(layout, (list, tuple)):
            layouts = list(layout)
            layout = layouts[0]
        if isinstance(layout, str):
            lay = get_layout(layout)
            if lay is None:
                raise FileNotFoundError(layout)
            layout = lay
        assert isinstance(layout, Layout)
        self.layout = layout
        self.layouts = layouts if layouts is not None else [layout]
        if any((l.width, l.height) != (layout.width, layout.height) for l in self.layouts):
            raise ValueError("all layouts of one PmxVecEnv must have the same width and height")
        self.n_envs = int(n_envs)
        self.length = int(length)
        self.device = torch.device(device)
        self.obs_code, self.obs_torch_dtype = _DTYPES[obs_dtype]
        self.obs_agents = tuple(sorted(set(int(a) for a in obs_agents)))
        cfg = _lib.Config()
        cfg.width, cfg.height = layout.width, layout.height
        L = self.layouts
        self._keep = (np.ascontiguousarray(np.stack([l.wall_rows for l in L])), np.ascontiguousarray(np.stack([l.food_rows for l in L])),
                      np.ascontiguousarray(np.stack([l.cap_rows for l in L])), np.ascontiguousarray(np.stack([l.starts for l in L])))
        if len(L) > 1:
            if layout_index is None:
                layout_index = np.arange(int(n_envs)) % len(L)
            self.layout_index = np.ascontiguousarray(layout_index, np.int32)
            assert self.layout_index.shape == (int(n_envs),)
            cfg.n_layouts = len(L)
            cfg.layout_index = self.layout_index.ctypes.data_as(C.POINTER(C.c_int32))
        else:
            self.layout_index = None
        cfg.wall_rows = self._keep[0].ctypes.data_as(C.POINTER(C.c_uint32))
        cfg.food_rows = self._keep[1].ctypes.data_as(C.POINTER(C.c_uint32))
        cfg.cap_rows = self._keep[2].ctypes.data_as(C.POINTER(C.c_uint32))
        cfg.starts = self._keep[3].ctypes.data_as(C.POINTER(C.c_int8))
        cfg.n_envs, cfg.length = self.n_envs, self.length
        cfg.legal_reward, cfg.defence_reward = int(bool(reward_forLegalAction)), int(bool(defenceReward))
        cfg.auto_reset = int(bool(auto_reset))
        cfg.obs_dtype = self.obs_code
        cfg.obs_agents = sum(1 << a for a in self.obs_agents)
        cfg.device = self.device.index or 0
        cfg.seed = int(seed) & 0xFFFFFFFF
        # random_layout=True of the reference (gymPacMan.py:98-100): an env moves to a freshly drawn layout of the pool at every
        # reset; `layout` is then the pool of generated mazes
        cfg.redraw_layouts = int(bool(redraw_layouts))
        if redraw_layouts and len(L) < 2:
            raise ValueError("redraw_layouts needs a pool of at least two layouts")
        cfg.enable_bots = int(bool(bots))          # understand ACTION_BASELINE_OFFENSE / _DEFENSE (in-kernel baselineTeam)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_create(C.byref(cfg), C.byref(self.handle)), "pmx_create")
        N, H, W = self.n_envs, layout.height, layout.width
        self.n_emit = len(self.obs_agents)
        self.obs_shape = (8, H, W)
        dev = self.device
        self.obs = torch.empty((N, self.n_emit, 8, H, W), dtype=self.obs_torch_dtype, device=dev)
        self.reward = torch.empty((N, 2), dtype=torch.float64, device=dev)
        self.done = torch.empty((N,), dtype=torch.uint8, device=dev)
        self.legal = torch.empty((N, 4), dtype=torch.uint8, device=dev)
        self.score_change = torch.empty((N,), dtype=torch.int32, device=dev)
        self.score = torch.empty((N,), dtype=torch.int32, device=dev)
        self.agent = torch.zeros((N, 4), dtype=torch.int32, device=dev)   # x | y<<8 | carry<<16 after own sub-step
        self._agent_obs = None

    # -- plumbing -------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _out(self, obs=True, obs_tensor=None):
        o = _lib.StepOut()
        t = obs_tensor if obs_tensor is not None else self.obs
        o.obs_dev = t.data_ptr() if obs else None
        o.reward_dev = self.reward.data_ptr()
        o.done_dev = self.done.data_ptr()
        o.legal_dev = self.legal.data_ptr()
        o.score_change_dev = self.score_change.data_ptr()
        o.score_dev = self.score.data_ptr()
        o.agent_dev = self.agent.data_ptr()
        return o

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.pmx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- gymPacMan surface, batched -----------------------------------------------------------------------------
    def reset(self, mask=None, want_obs=True):
        """gymPacMan.reset (gymPacMan.py:92-141).  mask: uint8/bool [N] tensor of envs to reset, None = all.
        Returns (obs [N,n_emit,8,H,W], legal [N,4] bit masks)."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert m.shape == (self.n_envs,)
        out = self._out(obs=want_obs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_reset(self.handle, m.data_ptr() if m is not None else None, C.byref(out),
                                          self._stream()), "pmx_reset")
        return self.obs, self.legal

    def step(self, actions, want_obs=True):
        """gymPacMan.step (gymPacMan.py:143-193) for all envs.  actions: int8 [N,4] (0 N, 1 E, 2 S, 3 W, 4 Stop).
        Returns (obs, reward [N,2] f64 (red, blue), done [N] u8, info dict of tensors).  The returned tensors are
        the env's own output buffers and are overwritten by the next call."""
        a = actions
        if a.dtype != torch.int8 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.int8).contiguous()
        assert a.shape == (self.n_envs, 4)
        out = self._out(obs=want_obs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_step(self.handle, a.data_ptr(), C.byref(out), self._stream()), "pmx_step")
        return self.obs, self.reward, self.done, {"legal_actions": self.legal, "score_change": self.score_change,
                                                   "score": self.score, "agent": self.agent}

    def step_agent(self, agent, actions, want_obs=True):
        """One agent's sub-step (loop body gymPacMan.py:149-169); agent 3 closes the tick.  actions int8 [N].
        Returns that agent's observation [N,8,H,W] (or None)."""
        a = actions.to(device=self.device, dtype=torch.int8).contiguous()
        assert a.shape == (self.n_envs,)
        if self._agent_obs is None:
            self._agent_obs = torch.empty((4, self.n_envs) + self.obs_shape, dtype=self.obs_torch_dtype, device=self.device)
        out = self._out(obs=want_obs, obs_tensor=self._agent_obs[agent])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_step_agent(self.handle, int(agent), a.data_ptr(), C.byref(out), self._stream()),
                       "pmx_step_agent")
        return self._agent_obs[agent] if want_obs else None

    def successor(self, agent, actions):
        """GameState.generateSuccessor (capture.py:107-123) in place on every env; returns scoreChange [N] int32."""
        a = actions.to(device=self.device, dtype=torch.int8).contiguous()
        assert a.shape == (self.n_envs,)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_successor(self.handle, int(agent), a.data_ptr(), self.score_change.data_ptr(),
                                              self._stream()), "pmx_successor")
        return self.score_change

    def observe(self, want_obs=True, want_legal=True):
        """get_Observation of the current state for every emitted agent (gymPacMan.py:195-229) and/or the legal masks."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_observe(self.handle, self.obs.data_ptr() if want_obs else None,
                                            self.legal.data_ptr() if want_legal else None, self._stream()), "pmx_observe")
        return self.obs, self.legal

    def emit_team_obs(self, team_red, team_obs, merged=None):
        """The two observations of one team for the CURRENT tick (canonicalised for red) into team_obs [N,2,8,H,W] and, if
        given, the merged critic input into merged [N,8,H,W] (pmx_emit_team_obs; pacman_mappo_resnet.py:215-229, :267-274)."""
        N = self.n_envs
        assert team_obs.shape == (N, 2) + self.obs_shape and team_obs.dtype == self.obs_torch_dtype and team_obs.is_contiguous()
        if merged is not None:
            assert merged.shape == (N,) + self.obs_shape and merged.dtype == self.obs_torch_dtype and merged.is_contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_emit_team_obs(self.handle, int(bool(team_red)), team_obs.data_ptr(),
                                                  merged.data_ptr() if merged is not None else None, self._stream()), "pmx_emit_team_obs")
        return team_obs, merged

    # -- measurement ---------------------------------------------------------------------------------------------
    def profile_begin(self, max_launches):
        _lib.check(self.lib.pmx_profile_begin(self.handle, int(max_launches)), "pmx_profile_begin")

    def set_tuning(self, key, value):
        """Launch tuning of the expansion kernel for A/B measurements (pmx_set_tuning): "expand_alt", "expand_nt", ..."""
        _lib.check(self.lib.pmx_set_tuning(self.handle, key.encode(), int(value)), "pmx_set_tuning")

    def profile_end(self):
        """-> dict(rule_ms, rule_launches, expand_ms, expand_launches): summed kernel times from HIP events."""
        rm, em, rn, en = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.pmx_profile_end(self.handle, C.byref(rm), C.byref(rn), C.byref(em), C.byref(en)), "pmx_profile_end")
        return dict(rule_ms=rm.value, rule_launches=rn.value, expand_ms=em.value, expand_launches=en.value)

    # -- state exchange ------------------------------------------------------------------------------------------
    def get_state(self, first=0, count=None):
        count = self.n_envs - first if count is None else count
        arr = (_lib.State * count)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_get_state(self.handle, first, count, arr, self._stream()), "pmx_get_state")
        return arr

    def set_state(self, states, first=0):
        count = len(states)
        arr = states if isinstance(states, C.Array) else (_lib.State * count)(*states)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_set_state(self.handle, first, count, arr, self._stream()), "pmx_set_state")

    def layout_indices(self):
        """The layout of the pool each env is on right now (int32 numpy array [n_envs])."""
        out = np.zeros(self.n_envs, np.int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_get_layout_index(self.handle, out.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()),
                       "pmx_get_layout_index")
        return out

    def maze_distances(self, layout=0):
        """distanceCalculator.computeDistances for a layout -> (cells [n,2] int8, dist [n,n] uint8) on the GPU."""
        n = C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmx_maze_distances_layout(self.handle, layout, None, None, C.byref(n), self._stream()),
                       "pmx_maze_distances")
            cells = torch.empty((n.value, 2), dtype=torch.int8, device=self.device)
            dist = torch.empty((n.value, n.value), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.pmx_maze_distances_layout(self.handle, layout, cells.data_ptr(), dist.data_ptr(), C.byref(n),
                                                          self._stream()), "pmx_maze_distances")
        return cells, dist


def make_state(pos, dir, pac, scared, carry, ret, food, caps, score, steps, H):
    s = _lib.State()
    for i in range(4):
        s.pos[i][0], s.pos[i][1] = int(pos[i][0]), int(pos[i][1])
        s.dir[i], s.pac[i], s.scared[i] = int(dir[i]), int(pac[i]), int(scared[i])
        s.carry[i], s.ret[i] = int(carry[i]), int(ret[i])
    for y in range(H):
        s.food[y], s.caps[y] = int(food[y]), int(caps[y])
    s.score, s.steps = int(score), int(steps)
    return s
