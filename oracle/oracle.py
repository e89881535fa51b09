"""ctypes front end of the CPU oracle (oracle/pmx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the product package.  Parity status: pinned against tests/golden (see the C file).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpmx_oracle.so")
MAXD = 32


def build(force=False):
    src = os.path.join(_HERE, "pmx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpmx_oracle.so"])
    return _LIB_PATH


class PState(C.Structure):
    _fields_ = [("pos", (C.c_int8 * 2) * 4), ("dir", C.c_int8 * 4), ("pac", C.c_uint8 * 4), ("scared", C.c_uint8 * 4),
                ("carry", C.c_uint8 * 4), ("ret", C.c_uint8 * 4), ("food", C.c_uint32 * MAXD), ("caps", C.c_uint32 * MAXD),
                ("score", C.c_int32), ("steps", C.c_int32), ("ticks", C.c_uint32)]


class SubOut(C.Structure):
    _fields_ = [("score_change", C.c_int32), ("win", C.c_int32), ("fault", C.c_int32), ("applied_action", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [("length", C.c_int), ("legal_reward", C.c_int), ("defence_reward", C.c_int), ("seed", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_sizeof_layout.restype = C.c_size_t
        _lib.orc_sizeof_state.restype = C.c_size_t
        _lib.orc_sizeof_pstate.restype = C.c_size_t
        _lib.orc_shaping.restype = C.c_double
        _lib.orc_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                 C.c_void_p, C.c_void_p]
        assert _lib.orc_sizeof_pstate() == C.sizeof(PState)
    return _lib


def parse_layout_text(rows):
    """layout.py:95-130: text rows (top first) -> W, H, wall/food/capsule bit rows (bottom-up), starts[4][2]."""
    H, W = len(rows), len(rows[0])
    walls = np.zeros(H, np.uint32)
    food = np.zeros(H, np.uint32)
    caps = np.zeros(H, np.uint32)
    agents = []
    for y in range(H):
        for x in range(W):
            ch = rows[H - 1 - y][x]
            if ch == "%":
                walls[y] |= np.uint32(1 << x)
            elif ch == ".":
                food[y] |= np.uint32(1 << x)
            elif ch == "o":
                caps[y] |= np.uint32(1 << x)
            elif ch in "1234":
                agents.append((int(ch), (x, y)))
    agents.sort()
    starts = np.array([p for _, p in agents], np.int8)
    return W, H, walls, food, caps, starts


class Layout:
    def __init__(self, rows):
        L = lib()
        self.rows = list(rows)
        self.W, self.H, self.walls, self.food, self.caps, self.starts = parse_layout_text(rows)
        assert self.starts.shape == (4, 2), "layout needs the four agent digits 1-4"
        self.buf = C.create_string_buffer(L.orc_sizeof_layout())
        rc = L.orc_layout_init(self.buf, self.W, self.H, self.walls.ctypes, self.food.ctypes, self.caps.ctypes,
                               self.starts.ctypes)
        assert rc == 0


class Env:
    """One oracle environment (the reference's gymPacMan_parallel_env with self_play=True)."""

    def __init__(self, rows, length=299, legal_reward=True, defence_reward=True):
        self.L = Layout(rows)
        self.lib = lib()
        self.cfg = Cfg(int(length), int(bool(legal_reward)), int(bool(defence_reward)), 0)
        self.S = C.create_string_buffer(self.lib.orc_sizeof_state())
        self.reset()

    @property
    def shape(self):
        return (8, self.L.H, self.L.W)

    def reset(self):
        self.lib.orc_reset(self.L.buf, self.S)

    def get_state(self):
        p = PState()
        self.lib.orc_pack(self.L.buf, self.S, C.byref(p))
        return p

    def set_state(self, p):
        self.lib.orc_unpack(self.L.buf, C.byref(p), self.S)

    def set_state_arrays(self, pos, dir, pac, scared, carry, ret, food, caps, score, steps):
        p = PState()
        for i in range(4):
            p.pos[i][0], p.pos[i][1] = int(pos[i][0]), int(pos[i][1])
            p.dir[i], p.pac[i], p.scared[i] = int(dir[i]), int(pac[i]), int(scared[i])
            p.carry[i], p.ret[i] = int(carry[i]), int(ret[i])
        for y in range(self.L.H):
            p.food[y], p.caps[y] = int(food[y]), int(caps[y])
        p.score, p.steps = int(score), int(steps)
        self.set_state(p)

    def legal(self, agent):
        return self.lib.orc_legal(self.L.buf, self.S, agent)

    def legal_list(self, agent):
        out = (C.c_int * 5)()
        n = self.lib.orc_legal_list(self.L.buf, self.S, agent, out)
        return [out[i] for i in range(n)]

    def obs(self, agent):
        o = np.zeros(self.shape, np.float32)
        self.lib.orc_encode_obs(self.L.buf, self.S, agent, o.ctypes)
        return o

    def substep(self, agent, action):
        so = SubOut()
        self.lib.orc_substep(self.L.buf, self.S, agent, int(action), C.byref(so))
        return so

    def tick(self, actions, want_sub=True):
        a = np.asarray(actions, np.int8)
        obs = np.zeros((4,) + self.shape, np.float32)
        reward = np.zeros(2, np.float64)
        done = C.c_uint8()
        legal = np.zeros(4, np.uint8)
        sc = C.c_int32()
        sub = (PState * 4)()
        so = (SubOut * 4)()
        self.lib.orc_tick(self.L.buf, C.byref(self.cfg), self.S, a.ctypes, obs.ctypes, reward.ctypes, C.byref(done),
                          legal.ctypes, C.byref(sc), sub, so)
        return dict(obs=obs, reward=reward, done=int(done.value), legal=legal, score_change=int(sc.value), sub=sub,
                    subout=so)


class BatchEnv:
    """N oracle envs of one layout stepped in a C loop (differential tests at scale, CPU baseline)."""

    def __init__(self, rows, n, length=299, legal_reward=True, defence_reward=True, auto_reset=True, seed=0):
        self.L = Layout(rows)
        self.lib = lib()
        self.n = n
        self.cfg = Cfg(int(length), int(bool(legal_reward)), int(bool(defence_reward)), int(seed))
        self.ssz = self.lib.orc_sizeof_state()
        self.S = C.create_string_buffer(self.ssz * n)
        self.auto_reset = int(auto_reset)
        for e in range(n):
            self.lib.orc_reset(self.L.buf, C.byref(self.S, e * self.ssz))
        self.reward = np.zeros((n, 2), np.float64)
        self.done = np.zeros(n, np.uint8)
        self.legal = np.zeros((n, 4), np.uint8)
        self.score_change = np.zeros(n, np.int32)
        self.score = np.zeros(n, np.int32)
        self.agent = np.zeros((n, 4), np.uint32)

    def tick(self, actions, obs=None):
        a = np.ascontiguousarray(actions, np.int8)
        assert a.shape == (self.n, 4)
        self.lib.orc_tick_batch(self.L.buf, C.byref(self.cfg), self.S, self.n, a.ctypes,
                                obs.ctypes if obs is not None else None, self.reward.ctypes, self.done.ctypes,
                                self.legal.ctypes, self.score_change.ctypes, self.auto_reset, self.score.ctypes,
                                self.agent.ctypes)

    def get_state(self, e):
        p = PState()
        self.lib.orc_pack(self.L.buf, C.byref(self.S, e * self.ssz), C.byref(p))
        return p

    def set_state(self, e, p):
        self.lib.orc_unpack(self.L.buf, C.byref(p), C.byref(self.S, e * self.ssz))


class MultiBatchEnv:
    """N oracle envs, each with its own (equally sized) layout."""

    def __init__(self, rows_list, layout_index, length=299, legal_reward=True, defence_reward=True, auto_reset=True, seed=0,
                 redraw=False):
        self.lib = lib()
        self.redraw = bool(redraw)      # move an env to a freshly drawn layout of the pool at every reset (gymPacMan.py:98-100)
        self.Ls = [Layout(r) for r in rows_list]
        lsz = self.lib.orc_sizeof_layout()
        self.lbuf = C.create_string_buffer(lsz * len(self.Ls))
        for k, L in enumerate(self.Ls):
            C.memmove(C.byref(self.lbuf, k * lsz), L.buf, lsz)
        self.index = np.ascontiguousarray(layout_index, np.int32)
        self.n = n = len(self.index)
        self.H, self.W = self.Ls[0].H, self.Ls[0].W
        self.cfg = Cfg(int(length), int(bool(legal_reward)), int(bool(defence_reward)), int(seed))
        self.ssz = self.lib.orc_sizeof_state()
        self.S = C.create_string_buffer(self.ssz * n)
        self.auto_reset = int(auto_reset)
        for e in range(n):
            self.lib.orc_reset(C.byref(self.lbuf, int(self.index[e]) * lsz), C.byref(self.S, e * self.ssz))
        self.reward = np.zeros((n, 2), np.float64)
        self.done = np.zeros(n, np.uint8)
        self.legal = np.zeros((n, 4), np.uint8)
        self.score_change = np.zeros(n, np.int32)
        self.score = np.zeros(n, np.int32)
        self.agent = np.zeros((n, 4), np.uint32)

    def tick(self, actions, obs=None):
        a = np.ascontiguousarray(actions, np.int8)
        self.lib.orc_tick_batch_multi(self.lbuf, self.index.ctypes, C.byref(self.cfg), self.S, self.n, a.ctypes,
                                      obs.ctypes if obs is not None else None, self.reward.ctypes, self.done.ctypes,
                                      self.legal.ctypes, self.score_change.ctypes, self.auto_reset, self.score.ctypes,
                                      self.agent.ctypes, len(self.Ls) if self.redraw else 0)


def bot_tables(rows):
    """(dist [n,n] uint8, cell_index int16[32*32], n) for the in-kernel baselineTeam action codes."""
    cells, dist = maze_distances(rows)
    idx = np.full(MAXD * MAXD, -1, np.int16)
    for k, (x, y) in enumerate(cells):
        idx[int(y) * MAXD + int(x)] = k
    return np.ascontiguousarray(dist), idx, len(cells)


def bot_best(env, agent, defensive, tables):
    """-> (bit mask of the best actions, home action or -1) for the reflex bot of `agent` on env's current state."""
    dist, idx, n = tables
    home = C.c_int(-1)
    m = lib().orc_bot_best(env.L.buf, env.S, int(agent), int(bool(defensive)), dist.ctypes, idx.ctypes, n, C.byref(home))
    return m, home.value


_bot_keep = None


def set_bot_tables(tables):
    """Install the distance tables the oracle uses for action codes -3 / -4 (the C side keeps raw pointers, so the arrays
    are held here for as long as they are installed)."""
    global _bot_keep
    _bot_keep = tables
    dist, idx, n = tables
    lib().orc_set_bot_tables(dist.ctypes, idx.ctypes, n)


def maze_distances(rows):
    L = Layout(rows)
    cells = np.zeros((MAXD * MAXD, 2), np.int8)
    n = lib().orc_maze_distances(L.buf, cells.ctypes, None, MAXD * MAXD)
    out = np.zeros((n, n), np.uint8)
    lib().orc_maze_distances(L.buf, cells.ctypes, out.ctypes, MAXD * MAXD)
    return cells[:n].copy(), out


def gae(rew, val, done, last_value, gamma=0.99, lam=0.95):
    rew = np.ascontiguousarray(rew, np.float32)
    val = np.ascontiguousarray(val, np.float32)
    done = np.ascontiguousarray(done, np.float32)
    adv = np.zeros_like(rew)
    ret = np.zeros_like(rew)
    lib().orc_gae(rew.ctypes.data, val.ctypes.data, done.ctypes.data, float(last_value), float(gamma), float(lam),
                  len(rew), adv.ctypes.data, ret.ctypes.data)
    return adv, ret


def canonicalize_obs(o):
    o = np.ascontiguousarray(o, np.float32)
    out = np.zeros_like(o)
    lib().orc_canonicalize_obs(o.ctypes, out.ctypes, o.shape[1], o.shape[2])
    return out


def merge_obs(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    lib().orc_merge_obs(a.ctypes, b.ctypes, out.ctypes, a.shape[1], a.shape[2])
    return out


def shaping(cur, nxt):
    cur = np.ascontiguousarray(cur, np.float32)
    nxt = np.ascontiguousarray(nxt, np.float32)
    return lib().orc_shaping(cur.ctypes, nxt.ctypes, cur.shape[1], cur.shape[2])


def dump_order(R):
    out = np.zeros(((2 * R + 1) ** 2, 2), np.int8)
    n = lib().orc_dump_order(R, out.ctypes, len(out))
    return out[:n].copy()
