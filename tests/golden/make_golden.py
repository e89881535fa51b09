#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Nothing of the reference is copied: this script drives the
reference's own Python objects with seeded inputs and records inputs + outputs
as small .npz/.json data files.  Those files are what pins the oracle
(oracle/pmx_oracle.c) and, through it, the HIP path.

Usage (from anywhere):
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py [section ...]
sections: traj scen dist maze gae ppo ppo_init ppo_init_blox ppo_init_tiny shaping bots   (default: all)

Fixture catalogue:
  G1 traj_*.npz      gymPacMan_parallel_env(self_play=True) trajectories, per sub-step state
  G2 (inside G1/G3)  get_Observation planes as u8 for every tick, canonicalize/merge samples
  G3 scen_*.npz      one-tick successors of hand-built and randomised states
  G4 dist_*.npz      distanceCalculator.computeDistances matrices
  G5 mazes.json      mazeGenerator.generateMaze(seed) text
  G6 gae.npz         pacman_mappo_resnet.compute_gae
  G7 ppo.npz         MAPPOAgent forward / PPO loss / grad-norm / Adam step with closed-form weights
  G7b ppo_init.npz   the same on the reference's own (seeded) orthogonal initialisation, paired rows: pins the bf16 production path
  G7c ppo_init_blox.npz  G7b on bloxCapture (20 x 20, the layout the reference trains on): pins the 28-tile tower and 416-token attention kernels
  G7d ppo_init_tiny.npz  G7b on tinyCapture (7 x 20, BASELINE configs[1]): the 10-tile tower kernels
  G8 shaping.npz     compute_heuristic_shaping / canonicalize_obs / merge_obs_for_critic
  G9 bots_*.json     baselineTeam / randomTeam action traces under random.seed(k)
"""
import contextlib
import io
import json
import os
import random
import signal
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF):
    sys.exit("reference not present: fixtures can only be regenerated in the build container")
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
os.chdir(os.environ.get("TMPDIR", "/tmp"))

with contextlib.redirect_stdout(io.StringIO()):
    import torch
    import capture
    import game
    import gymPacMan
    import layout as ref_layout
    import distanceCalculator
    import mazeGenerator

LAYOUT_FILES = {
    "tiny": f"{REF}/layouts/tinyCapture.lay",
    "small": f"{REF}/layouts/smallCapture.lay",
    "blox": f"{REF}/layouts/bloxCapture.lay",
}
DIR2INT = {"North": 0, "East": 1, "South": 2, "West": 3, "Stop": 4}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def layout_text(name):
    """Layout rows (top row first) for a file layout or 'maze<seed>'."""
    if name.startswith("maze"):
        seed = int(name[4:])
        st = random.getstate()
        txt = quiet(mazeGenerator.generateMaze, seed)
        random.setstate(st)
        return txt.split("\n")
    with open(LAYOUT_FILES[name]) as f:
        return [ln.strip() for ln in f]


def make_env(name, length, self_play=True, enemy="randomTeam", legal_reward=True, defence=True):
    if name.startswith("maze"):
        # the env has no "layout from text" entry: write the maze to a scratch .lay file
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pmx_{name}.lay")
        with open(path, "w") as f:
            f.write("\n".join(layout_text(name)))
    else:
        path = LAYOUT_FILES[name]
    env = quiet(gymPacMan.gymPacMan_parallel_env, layout_file=path, length=length,
                reward_forLegalAction=legal_reward, defenceReward=defence,
                enemieName=enemy, self_play=self_play)
    return env


def grid_rows(grid):
    """Grid (x-major list of lists) -> uint32[H], bit x of row y."""
    rows = np.zeros(grid.height, dtype=np.uint32)
    for x in range(grid.width):
        col = grid.data[x]
        for y in range(grid.height):
            if col[y]:
                rows[y] |= np.uint32(1 << x)
    return rows


def caps_rows(caps, H):
    rows = np.zeros(H, dtype=np.uint32)
    for (x, y) in caps:
        rows[int(y)] |= np.uint32(1 << int(x))
    return rows


def snapshot(state):
    """All dynamic fields of a reference GameState as plain arrays."""
    d = state.data
    H = d.layout.height
    pos = np.zeros((4, 2), np.int8)
    dirs = np.zeros(4, np.int8)
    pac = np.zeros(4, np.uint8)
    scared = np.zeros(4, np.uint8)
    carry = np.zeros(4, np.uint8)
    ret = np.zeros(4, np.uint8)
    for i, a in enumerate(d.agentStates):
        x, y = a.configuration.pos
        assert x == int(x) and y == int(y)
        pos[i] = (int(x), int(y))
        dirs[i] = DIR2INT[a.configuration.direction]
        pac[i] = bool(a.isPacman)
        scared[i] = a.scaredTimer
        carry[i] = a.numCarrying
        ret[i] = a.numReturned
    return dict(pos=pos, dir=dirs, pac=pac, scared=scared, carry=carry, ret=ret,
                food=grid_rows(d.food), caps=caps_rows(d.capsules, H),
                score=np.int32(d.score), schange=np.int32(d.scoreChange), win=np.uint8(bool(d._win)))


SNAP_KEYS = ("pos", "dir", "pac", "scared", "carry", "ret", "food", "caps", "score", "schange", "win")


def legal_mask(lst):
    m = 0
    for a in lst:
        m |= 1 << int(a)
    return m


def obs_u8(t):
    a = t.numpy()
    assert a.dtype == np.float32
    b = a.astype(np.uint8)
    assert (b.astype(np.float32) == a).all()
    return b


class SubstepRecorder:
    """Hooks env.get_Observation: gymPacMan.step calls it right after each agent's sub-step
    (gymPacMan.py:166), when env.game.state is the post-sub-step state."""

    def __init__(self, env):
        self.env = env
        self.orig = env.get_Observation
        self.snaps = []
        env.get_Observation = self

    def __call__(self, idx):
        self.snaps.append(snapshot(self.env.game.state))
        return self.orig(idx)


def tick_record(env, rec, actions):
    """One env.step -> dict of everything the tick produced."""
    rec.snaps.clear()
    obs, rew, term, info = env.step({env.agents[i]: int(actions[i]) for i in range(4)})
    assert len(rec.snaps) == 4
    out = {"sub_" + k: np.stack([s[k] for s in rec.snaps]) for k in SNAP_KEYS}
    out["obs"] = np.stack([obs_u8(obs[env.agents[i]]) for i in range(4)])
    out["reward"] = np.array([rew[env.agents[0]], rew[env.agents[1]]], np.float64)
    assert rew[env.agents[2]] == rew[env.agents[0]] and rew[env.agents[3]] == rew[env.agents[1]]
    out["done"] = np.uint8(any(term.values()))
    out["legal"] = np.array([legal_mask(info["legal_actions"][env.agents[i]]) for i in range(4)], np.uint8)
    out["legal_lists"] = [list(map(int, info["legal_actions"][env.agents[i]])) for i in range(4)]
    out["score_change"] = np.int32(info["score_change"])
    return out


# --------------------------------------------------------------------------------------
# action policies for G1
# --------------------------------------------------------------------------------------
class Hunter:
    """Scripted policy that makes things happen: eat, carry home, chase invaders.
    Uses the reference Distancer only to choose inputs; the inputs themselves are recorded."""

    def __init__(self, env, rng, eps):
        self.env, self.rng, self.eps = env, rng, eps
        self.dist = distanceCalculator.Distancer(env.layout)
        quiet(self.dist.getMazeDistances)
        self.quota = [int(rng.randint(1, 6)) for _ in range(4)]

    def act(self, i, legal):
        st = self.env.game.state
        if self.rng.rand() < self.eps:
            return int(self.rng.randint(5))
        me = st.getAgentState(i)
        pos = st.getAgentPosition(i)
        red = i in (0, 2)
        W = self.env.layout.width
        enemies = [j for j in range(4) if (j in (0, 2)) != red]
        target = None
        inv = [j for j in enemies if st.getAgentState(j).isPacman]
        if i in (2, 3) and inv and me.scaredTimer == 0:
            target = [st.getAgentPosition(j) for j in inv]
        elif me.numCarrying >= self.quota[i]:
            hx = W // 2 - 1 if red else W // 2
            target = [(hx, y) for y in range(self.env.layout.height) if not self.env.layout.walls[hx][y]]
        else:
            food = (st.getBlueFood() if red else st.getRedFood()).asList()
            target = food if food else None
        if not target:
            return int(self.rng.choice(legal))
        best, bestd = [], 10 ** 9
        for a in legal:
            dx, dy = game.Actions.directionToVector(self.env.action_mapping[a])
            np_ = (int(pos[0] + dx), int(pos[1] + dy))
            try:
                d = min(self.dist.getDistance(np_, t) for t in target)
            except Exception:
                d = 10 ** 8
            if d < bestd:
                best, bestd = [a], d
            elif d == bestd:
                best.append(a)
        return int(self.rng.choice(best))


def gen_traj(tag, lay, length, policy, seed, ticks, eps=0.15, legal_reward=True, defence=True):
    rng = np.random.RandomState(seed)
    env = make_env(lay, length, legal_reward=legal_reward, defence=defence)
    rec = SubstepRecorder(env)
    T = ticks
    cols = {}
    resets = np.zeros(T, np.uint8)
    actions = np.zeros((T, 4), np.int8)
    legal_lists_sample = []
    obs0, info0 = quiet(env.reset)
    init_obs = np.stack([obs_u8(obs0[i]) for i in range(4)])
    init_legal = np.array([legal_mask(info0["legal_actions"][i]) for i in range(4)], np.uint8)
    init_snap = snapshot(env.game.state)
    legal = [info0["legal_actions"][i] for i in range(4)]
    hunter = Hunter(env, rng, eps) if policy == "hunter" else None
    events = dict(eats=0, returns=0, deaths=0, dumps=0, caps=0, dones=0)
    prev = init_snap
    for t in range(T):
        if policy == "uniform":
            a = rng.randint(5, size=4)
        elif policy == "legal":
            a = [int(rng.choice(legal[i])) for i in range(4)]
        else:
            # the hunter looks at the state at the start of the tick for every agent
            a = [hunter.act(i, legal[i]) for i in range(4)]
        actions[t] = a
        r = tick_record(env, rec, a)
        for k, v in r.items():
            if k == "legal_lists":
                if t < 8:
                    legal_lists_sample.append(v)
                continue
            cols.setdefault(k, []).append(v)
        for s in range(4):
            cur = {k: r["sub_" + k][s] for k in SNAP_KEYS}
            if cur["carry"].sum() > prev["carry"].sum():
                events["eats"] += 1
            if cur["schange"] != 0:
                events["returns"] += 1
            if int(cur["caps"].sum()) < int(prev["caps"].sum()):
                events["caps"] += 1
            fc = sum(bin(int(x)).count("1") for x in cur["food"])
            fp = sum(bin(int(x)).count("1") for x in prev["food"])
            if fc > fp:
                events["dumps"] += 1
            for j in range(4):
                if prev["pac"][j] and not cur["pac"][j] and tuple(cur["pos"][j]) == tuple(init_snap["pos"][j]) \
                        and abs(int(cur["pos"][j][0]) - int(prev["pos"][j][0])) + abs(int(cur["pos"][j][1]) - int(prev["pos"][j][1])) > 1:
                    events["deaths"] += 1
            prev = cur
        legal = [[a_ for a_ in range(5) if (int(r["legal"][i]) >> a_) & 1] for i in range(4)]
        if r["done"]:
            events["dones"] += 1
            resets[t] = 1
            obs0, info0 = quiet(env.reset)
            assert (np.stack([obs_u8(obs0[i]) for i in range(4)]) == init_obs).all()
            legal = [info0["legal_actions"][i] for i in range(4)]
            prev = init_snap
            if hunter:
                hunter.quota = [int(rng.randint(1, 6)) for _ in range(4)]
    data = {k: np.stack(v) for k, v in cols.items()}
    data.update(actions=actions, resets=resets, init_obs=init_obs, init_legal=init_legal)
    data.update({"init_" + k: init_snap[k] for k in SNAP_KEYS})
    meta = dict(layout=layout_text(lay), layout_name=lay, length=length, policy=policy, seed=seed,
                legal_reward=legal_reward, defence=defence, events=events,
                legal_lists_first_ticks=legal_lists_sample,
                source="gymPacMan.gymPacMan_parallel_env(self_play=True).step")
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    path = os.path.join(OUT, f"traj_{tag}.npz")
    np.savez_compressed(path, **data)
    print(f"  {os.path.basename(path)}: T={T} events={events} size={os.path.getsize(path)}")


def section_traj():
    print("G1 trajectories")
    gen_traj("tiny_uniform", "tiny", 300, "uniform", 11, 400)
    gen_traj("tiny_hunter", "tiny", 300, "hunter", 12, 700)
    gen_traj("small_uniform", "small", 300, "uniform", 21, 400)
    gen_traj("small_legal", "small", 40, "legal", 22, 300)
    gen_traj("small_hunter", "small", 300, "hunter", 23, 900)
    gen_traj("small_hunter_noshape", "small", 299, "hunter", 24, 350, legal_reward=False, defence=False)
    gen_traj("blox_hunter", "blox", 300, "hunter", 31, 650)
    gen_traj("blox_uniform", "blox", 300, "uniform", 32, 200)
    gen_traj("maze23_hunter", "maze23", 300, "hunter", 41, 650)
    gen_traj("maze4242_hunter", "maze4242", 300, "hunter", 42, 650, eps=0.3)


# --------------------------------------------------------------------------------------
# G3 scenarios: hand-built / randomised states -> one tick
# --------------------------------------------------------------------------------------
INT2DIR = {v: k for k, v in DIR2INT.items()}


def inject(env, st):
    """Overwrite the dynamic part of env.game.state from a plain dict (same keys as snapshot())."""
    s = env.game.state
    d = s.data
    W, H = d.layout.width, d.layout.height
    food = game.Grid(W, H, False)
    for y in range(H):
        for x in range(W):
            if (int(st["food"][y]) >> x) & 1:
                food[x][y] = True
    d.food = food
    caps = []
    for y in range(H):
        for x in range(W):
            if (int(st["caps"][y]) >> x) & 1:
                caps.append((x, y))
    d.capsules = caps
    for i, a in enumerate(d.agentStates):
        p = (int(st["pos"][i][0]), int(st["pos"][i][1]))
        a.configuration = game.Configuration(p, INT2DIR[int(st["dir"][i])])
        a.isPacman = bool(st["pac"][i])
        a.scaredTimer = int(st["scared"][i])
        a.numCarrying = int(st["carry"][i])
        a.numReturned = int(st["ret"][i])
    d.score = int(st["score"])
    d.scoreChange = 0
    env.steps = int(st.get("steps", 0))


class _Timeout(Exception):
    pass


def _alarm(*_):
    raise _Timeout()


def run_scenarios(tag, lay, states, actions, length=300, legal_reward=True, defence=True):
    env = make_env(lay, length, legal_reward=legal_reward, defence=defence)
    rec = SubstepRecorder(env)
    quiet(env.reset)
    keep_states, keep_actions, cols, skipped = [], [], {}, 0
    signal.signal(signal.SIGALRM, _alarm)
    for st, a in zip(states, actions):
        inject(env, st)
        signal.alarm(2)
        try:
            r = tick_record(env, rec, a)
        except (_Timeout, Exception) as e:  # reference raises / never ends on states it cannot handle
            skipped += 1
            rec.snaps.clear()
            continue
        finally:
            signal.alarm(0)
        keep_states.append(st)
        keep_actions.append(a)
        for k, v in r.items():
            if k != "legal_lists":
                cols.setdefault(k, []).append(v)
    data = {k: np.stack(v) for k, v in cols.items()}
    for k in ("pos", "dir", "pac", "scared", "carry", "ret", "food", "caps", "score"):
        data["in_" + k] = np.stack([np.asarray(s[k]) for s in keep_states])
    data["in_steps"] = np.array([int(s.get("steps", 0)) for s in keep_states], np.int32)
    data["actions"] = np.array(keep_actions, np.int8)
    meta = dict(layout=layout_text(lay), layout_name=lay, length=length, legal_reward=legal_reward,
                defence=defence, skipped=skipped, names=[s.get("name", "") for s in keep_states],
                source="hand-built GameState -> gymPacMan.step")
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    path = os.path.join(OUT, f"scen_{tag}.npz")
    np.savez_compressed(path, **data)
    print(f"  {os.path.basename(path)}: K={len(keep_states)} skipped={skipped} size={os.path.getsize(path)}")


def base_state(env):
    s = snapshot(env.game.state)
    s = {k: np.array(v) for k, v in s.items()}
    s["steps"] = 0
    return s


def is_red_side(x, W):
    return x < W / 2


def fix_pac(st, W):
    for i in range(4):
        st["pac"][i] = (i in (0, 2)) != is_red_side(int(st["pos"][i][0]), W)


def random_states(lay, n, seed, with_caps):
    rng = np.random.RandomState(seed)
    env = make_env(lay, 300)
    quiet(env.reset)
    base = base_state(env)
    W, H = env.layout.width, env.layout.height
    walls = env.layout.walls
    open_cells = [(x, y) for x in range(W) for y in range(H) if not walls[x][y]]
    states, actions = [], []
    for k in range(n):
        st = {kk: np.array(v) for kk, v in base.items()}
        st["steps"] = int(rng.choice([0, 5, 299, 300, 301])) if rng.rand() < 0.1 else int(rng.randint(0, 300))
        # food: random subset of open cells, sometimes sparse, sometimes none on one side
        mode = rng.randint(5)
        p = [0.5, 0.15, 0.03, 0.9, 0.3][mode]
        food = np.zeros(H, np.uint32)
        for (x, y) in open_cells:
            if rng.rand() < p:
                food[y] |= np.uint32(1 << x)
        if rng.rand() < 0.15:  # clear one side completely (termination test)
            half = int(W / 2)
            mask = (1 << half) - 1
            if rng.rand() < 0.5:
                food &= np.uint32(mask)
            else:
                food &= np.uint32(~mask & 0xFFFFFFFF)
        caps = np.zeros(H, np.uint32)
        if with_caps:
            for _ in range(rng.randint(0, 4)):
                x, y = open_cells[rng.randint(len(open_cells))]
                if rng.rand() < 0.3:
                    x = int(W / 2) + int(rng.randint(-1, 2))
                    if walls[x][y]:
                        continue
                caps[y] |= np.uint32(1 << x)
                food[y] &= np.uint32(~(1 << x) & 0xFFFFFFFF)
        st["food"], st["caps"] = food, caps
        # agents: clustered so that collisions are common
        cx, cy = open_cells[rng.randint(len(open_cells))]
        for i in range(4):
            if rng.rand() < 0.15:
                continue  # stays on its start
            if rng.rand() < 0.7:
                near = [c for c in open_cells if abs(c[0] - cx) + abs(c[1] - cy) <= 2]
                x, y = near[rng.randint(len(near))]
            else:
                x, y = open_cells[rng.randint(len(open_cells))]
            st["pos"][i] = (x, y)
            st["dir"][i] = rng.randint(5)
        fix_pac(st, W)
        for i in range(4):
            st["scared"][i] = 0 if rng.rand() < 0.6 else int(rng.choice([1, 2, 39, 40, rng.randint(1, 41)]))
            st["carry"][i] = 0 if rng.rand() < 0.4 else int(rng.choice([1, 2, 3, 7, 15, rng.randint(1, 20)]))
            st["ret"][i] = int(rng.randint(0, 6))
            if not st["pac"][i] and rng.rand() < 0.8:
                st["carry"][i] = 0  # a ghost normally carries nothing (it returned); keep some odd ones
        # a pacman standing on food would already have eaten it: usually clear, sometimes keep (Q1/Q2 paths)
        for i in range(4):
            if rng.rand() < 0.7:
                x, y = int(st["pos"][i][0]), int(st["pos"][i][1])
                st["food"][y] &= np.uint32(~(1 << x) & 0xFFFFFFFF)
        st["score"] = np.int32(rng.randint(-8, 9))
        st["name"] = f"rand{k}"
        states.append(st)
        actions.append([int(rng.randint(5)) for _ in range(4)])
    return states, actions


def handmade_small():
    """Named scenarios on smallCapture (W=14, H=11; red side x<7).  Open rows: y=1,3,5,7,9 are corridors."""
    env = make_env("small", 300)
    quiet(env.reset)
    base = base_state(env)
    W = 14
    S, A = [], []

    def mk(name, **kw):
        st = {k: np.array(v) for k, v in base.items()}
        st["name"] = name
        for k, v in kw.items():
            if k == "agents":
                for i, spec in v.items():
                    st["pos"][i] = spec[0]
                    if len(spec) > 1:
                        st["carry"][i] = spec[1]
                    if len(spec) > 2:
                        st["scared"][i] = spec[2]
                fix_pac(st, W)
            elif k == "food_add":
                for (x, y) in v:
                    st["food"][y] |= np.uint32(1 << x)
            elif k == "food_clear":
                for (x, y) in v:
                    st["food"][y] &= np.uint32(~(1 << x) & 0xFFFFFFFF)
            elif k == "caps":
                for (x, y) in v:
                    st["caps"][y] |= np.uint32(1 << x)
                    st["food"][y] &= np.uint32(~(1 << x) & 0xFFFFFFFF)
            elif k == "pac":
                for i, p in v.items():
                    st["pac"][i] = p
            else:
                st[k] = v
        return st

    N, E, So, Wd, ST = 0, 1, 2, 3, 4
    # Q1: red agent 0 returns (blue side (7,5) -> red side (6,5)) onto own-side food while agent 3 is a Pacman
    S.append(mk("q1_return_eats_own_food_agent3_pacman", agents={0: ((7, 5), 3), 3: ((5, 7), 0)}, food_add=[(6, 5)]))
    A.append([Wd, ST, ST, ST])
    S.append(mk("q1_return_agent3_ghost_no_eat", agents={0: ((7, 5), 3), 3: ((9, 7), 0)}, food_add=[(6, 5)]))
    A.append([Wd, ST, ST, ST])
    # Q2: agent 2 returns onto a cell where agent 0 idles on own food, agent 3 Pacman -> agent 0 gets the carry
    S.append(mk("q2_teammate_gets_credit", agents={0: ((6, 5), 0), 2: ((7, 5), 2), 3: ((5, 7), 0)}, food_add=[(6, 5)]))
    A.append([ST, ST, Wd, ST])
    # blue mirror of Q1 (agent 3 itself returns; it is a ghost after the move -> no eat)
    S.append(mk("q1_blue_agent3_returns", agents={3: ((6, 5), 4)}, food_add=[(7, 5)]))
    A.append([ST, ST, ST, E])
    S.append(mk("q1_blue_agent1_returns_agent3_pacman", agents={1: ((6, 5), 4), 3: ((4, 5), 1)}, food_add=[(7, 5)]))
    A.append([ST, E, ST, ST])
    # Q3 capsules: x == W/2 = 7 counts as red's (blue eats it), red eats only x > 7
    S.append(mk("q3_capsule_on_boundary_red_steps_on", agents={0: ((6, 5), 0)}, caps=[(7, 5)]))
    A.append([E, ST, ST, ST])
    S.append(mk("q3_capsule_on_boundary_blue_steps_on", agents={1: ((8, 5), 0)}, caps=[(7, 5)]))
    A.append([ST, Wd, ST, ST])
    S.append(mk("capsule_red_eats_blue_capsule", agents={0: ((8, 5), 1), 1: ((10, 5), 0), 3: ((9, 7), 0)}, caps=[(9, 5)]))
    A.append([E, Wd, ST, ST])  # red eats capsule at (9,5); blue 1 (now scared) walks into it -> blue 1 goes home
    S.append(mk("capsule_blue_eats_red_capsule", agents={1: ((5, 5), 1), 0: ((3, 5), 0)}, caps=[(4, 5)]))
    A.append([E, Wd, ST, ST])
    # scared ghost mover steps onto pacman -> ghost sent home
    S.append(mk("scared_ghost_moves_onto_pacman", agents={0: ((8, 5), 2), 1: ((9, 5), 0, 5)}))
    A.append([ST, Wd, ST, ST])
    S.append(mk("scared_timer_one_expires", agents={1: ((9, 5), 0, 1), 0: ((8, 5), 2)}))
    A.append([ST, ST, ST, ST])
    # kills and dumps
    S.append(mk("ghost_kills_carrier_dump", agents={0: ((8, 5), 9), 1: ((9, 5), 0)}))
    A.append([ST, Wd, ST, ST])
    S.append(mk("pacman_suicide_dump_near_border", agents={0: ((12, 9), 12), 1: ((12, 8), 0)}))
    A.append([So, ST, ST, ST])
    S.append(mk("dump_big_carry_sparse_board", agents={2: ((9, 3), 25), 3: ((10, 3), 0)},
                food=np.zeros(11, np.uint32)))
    A.append([ST, ST, ST, Wd])
    S.append(mk("dump_blocked_by_agents_and_capsule", agents={1: ((4, 3), 6), 0: ((4, 4), 0), 2: ((3, 3), 0), 3: ((5, 3), 0)},
                caps=[(2, 3)]))
    A.append([So, ST, ST, ST])
    S.append(mk("two_ghosts_same_cell_pacman_moves_in", agents={0: ((8, 7), 4), 1: ((9, 7), 0), 3: ((9, 7), 0)}))
    A.append([E, ST, ST, ST])
    S.append(mk("two_ghosts_first_scared", agents={0: ((8, 7), 4), 1: ((9, 7), 0, 7), 3: ((9, 7), 0)}))
    A.append([E, ST, ST, ST])
    S.append(mk("ghost_moves_onto_two_pacmen", agents={1: ((4, 7), 3), 3: ((4, 7), 2), 0: ((3, 7), 0)}))
    A.append([E, ST, ST, ST])
    S.append(mk("scared_ghost_onto_two_pacmen", agents={1: ((4, 7), 3), 3: ((4, 7), 2), 0: ((3, 7), 0, 9)}))
    A.append([E, ST, ST, ST])
    # termination: blue-side food gone and red carries nothing -> done with bonus for the leader
    blue_clear = np.array(base["food"]) & np.uint32((1 << 7) - 1)
    S.append(mk("terminal_all_blue_food_gone_red_leads", food=blue_clear.copy(), score=np.int32(5)))
    A.append([ST, ST, ST, ST])
    S.append(mk("terminal_blue_food_gone_but_red_carrying", food=blue_clear.copy(), agents={0: ((9, 5), 2)}))
    A.append([ST, ST, ST, ST])
    S.append(mk("terminal_last_pellet_returned_this_tick", food=blue_clear.copy(), agents={0: ((7, 5), 14)}, score=np.int32(0)))
    A.append([Wd, ST, ST, ST])
    red_clear = np.array(base["food"]) & np.uint32(~((1 << 7) - 1) & 0xFFFFFFFF)
    S.append(mk("terminal_red_food_gone_blue_leads", food=red_clear.copy(), score=np.int32(-7)))
    A.append([ST, ST, ST, ST])
    S.append(mk("terminal_time_limit_tie", steps=300, score=np.int32(0)))
    A.append([ST, ST, ST, ST])
    S.append(mk("time_limit_minus_one", steps=299, score=np.int32(3)))
    A.append([ST, ST, ST, ST])
    S.append(mk("terminal_time_limit_red_leads", steps=300, score=np.int32(3)))
    A.append([N, N, N, N])
    # illegal actions into walls, all directions
    S.append(mk("illegal_moves_become_stop"))
    A.append([N, So, E, Wd])
    return S, A


def section_scen():
    print("G3 scenarios")
    S, A = handmade_small()
    run_scenarios("small_named", "small", S, A)
    for lay, n, seed, caps in (("small", 1200, 101, True), ("tiny", 600, 102, True), ("blox", 400, 103, False),
                               ("maze23", 500, 104, True)):
        S, A = random_states(lay, n, seed, caps)
        run_scenarios(f"{lay}_random", lay, S, A)


# --------------------------------------------------------------------------------------
def section_dist():
    print("G4 maze distances")
    for lay in ("tiny", "small", "blox", "maze23"):
        L = ref_layout.Layout(layout_text(lay))
        d = distanceCalculator.computeDistances(L)
        cells = L.walls.asList(False)
        n = len(cells)
        m = np.zeros((n, n), np.int64)
        for a, ca in enumerate(cells):
            for b, cb in enumerate(cells):
                m[a, b] = d[(ca, cb)]
        unreachable = m == sys.maxsize
        m[unreachable] = 255
        assert m.max() <= 255
        path = os.path.join(OUT, f"dist_{lay}.npz")
        np.savez_compressed(path, cells=np.array(cells, np.int8), dist=m.astype(np.uint8),
                            meta=np.frombuffer(json.dumps(dict(layout=layout_text(lay), unreachable_code=255,
                                                               source="distanceCalculator.computeDistances")).encode(), np.uint8))
        print(f"  {os.path.basename(path)}: n={n} max={int(m[~unreachable].max())} unreachable={int(unreachable.sum())}")


def section_maze():
    print("G5 mazes")
    seeds = list(range(1, 17)) + [23, 99, 1234, 4242, 9999, 65536]
    out = {}
    for s in seeds:
        out[str(s)] = quiet(mazeGenerator.generateMaze, s).split("\n")
    # capture.randomLayout(seed) is the env's entry point (capture.py:905-911)
    out["randomLayout_7"] = quiet(capture.randomLayout, 7).split("\n")
    with open(os.path.join(OUT, "mazes.json"), "w") as f:
        json.dump(dict(source="mazeGenerator.generateMaze(seed)", mazes=out), f, indent=0)
    print(f"  mazes.json: {len(out)} mazes")


def section_gae():
    print("G6 GAE")
    with contextlib.redirect_stdout(io.StringIO()):
        import pacman_mappo_resnet as M
    rng = np.random.RandomState(7)
    cases = {}
    for name, T, pdone in (("a", 2048, 1 / 301.0), ("b", 512, 0.05), ("c", 33, 0.0), ("d", 1, 1.0), ("e", 64, 0.5)):
        rew = torch.tensor(rng.randn(T).astype(np.float32) * (1 + 5 * (rng.rand(T) < 0.02)).astype(np.float32))
        val = torch.tensor(rng.randn(T).astype(np.float32))
        done = torch.tensor((rng.rand(T) < pdone).astype(np.float32))
        last = float(np.float32(rng.randn()))
        adv, ret = M.compute_gae(rew, val, done, last, M.GAMMA)
        cases[name + "_rew"], cases[name + "_val"], cases[name + "_done"] = rew.numpy(), val.numpy(), done.numpy()
        cases[name + "_last"] = np.float32(last)
        cases[name + "_adv"], cases[name + "_ret"] = adv.numpy(), ret.numpy()
    cases["meta"] = np.frombuffer(json.dumps(dict(gamma=M.GAMMA, lam=M.GAE_LAMBDA, cases=list("abcde"),
                                                  source="pacman_mappo_resnet.compute_gae")).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **cases)
    print("  gae.npz")


def closed_form_weights(model):
    """Deterministic weights with sane scale: p[k] = s * sin(0.37*k + 1.3*j) for the j-th tensor,
    s = 0.5/sqrt(fan_in) for matrices/filters, LayerNorm/GroupNorm weights 1 + 0.1*sin, biases 0.05*sin."""
    with torch.no_grad():
        for j, (name, p) in enumerate(model.named_parameters()):
            k = torch.arange(p.numel(), dtype=torch.float64)
            base = torch.sin(0.37 * k + 1.3 * j)
            if p.dim() >= 2:
                fan_in = p[0].numel()
                v = base * (0.5 / np.sqrt(fan_in))
            elif name.endswith("weight"):
                v = 1.0 + 0.1 * base
            else:
                v = 0.05 * base
            p.copy_(v.reshape(p.shape).to(torch.float32))


def section_ppo():
    print("G7 PPO losses")
    with contextlib.redirect_stdout(io.StringIO()):
        import pacman_mappo_resnet as M
    torch.manual_seed(0)
    torch.set_num_threads(1)
    z = np.load(os.path.join(OUT, "traj_small_hunter.npz"))
    obs_all = z["obs"]  # [T,4,8,H,W] u8
    B = 48
    idx = np.arange(10, 10 + B * 5, 5)
    ob1 = torch.tensor(obs_all[idx, 1].astype(np.float32))
    ob3 = torch.tensor(obs_all[idx, 3].astype(np.float32))
    obs = torch.cat([ob1[: B // 2], ob3[B // 2:]])
    merged = torch.stack([M.merge_obs_for_critic([ob1[i], ob3[i]]) for i in range(B)])
    rng = np.random.RandomState(3)
    act = torch.tensor(rng.randint(5, size=B), dtype=torch.long)
    old_logp = torch.tensor((np.log(0.2) + 0.3 * rng.randn(B)).astype(np.float32))
    adv = torch.tensor(rng.randn(B).astype(np.float32))
    ret = torch.tensor(rng.randn(B).astype(np.float32))
    model = M.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    closed_form_weights(model)
    clip_eps, ent_coef, lr = M.CLIP_EPS, 0.02, 2e-4
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-5)
    vals, lps, ent = model.evaluate(obs, merged, act)
    with torch.no_grad():
        logits = model.actor_head(model.actor_backbone(obs))
    norm_adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = (lps - old_logp).exp()
    pg = -torch.min(norm_adv * ratio, norm_adv * torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps)).mean()
    vl = 0.5 * ((vals - ret) ** 2).mean()
    loss = pg + M.VF_COEF * vl - ent_coef * ent.mean()
    opt.zero_grad()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), M.MAX_GRAD_NORM)
    opt.step()
    psum = sum(float(p.double().sum()) for p in model.parameters())
    pabs = sum(float(p.double().abs().sum()) for p in model.parameters())
    out = dict(obs=obs.numpy().astype(np.uint8), merged=merged.numpy().astype(np.uint8), act=act.numpy(),
               old_logp=old_logp.numpy(), adv=adv.numpy(), ret=ret.numpy(), logits=logits.numpy(),
               values=vals.detach().numpy(), logp=lps.detach().numpy(), entropy=ent.detach().numpy(),
               pg=np.float32(pg.item()), vl=np.float32(vl.item()), loss=np.float32(loss.item()),
               grad_norm=np.float32(float(gn)), post_adam_sum=np.float64(psum), post_adam_abs=np.float64(pabs),
               n_params=np.int64(sum(p.numel() for p in model.parameters())))
    out["meta"] = np.frombuffer(json.dumps(dict(
        clip_eps=clip_eps, ent_coef=ent_coef, lr=lr, vf_coef=M.VF_COEF, max_grad_norm=M.MAX_GRAD_NORM,
        weights="p_j[k]=s*sin(0.37k+1.3j); s=0.5/sqrt(fan_in) (dim>=2), 1+0.1*sin (norm weight), 0.05*sin (bias)",
        param_names=[n for n, _ in model.named_parameters()],
        source="pacman_mappo_resnet.MAPPOAgent.evaluate + PPO loss lines 577-590")).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "ppo.npz"), **out)
    print(f"  ppo.npz loss={loss.item():.6f} pg={pg.item():.6f} vl={vl.item():.6f} gn={float(gn):.6f}")


def sharpen_init(model):
    """A deterministic transform of an initialised MAPPOAgent that both sides can apply (tests/test_mappo_cpu.py has the same
    function): the reference's init makes near-uniform policies (last actor layer: gain 0.01) and near-constant values, which pin
    little; scaled up, logits and values spread over O(1) while every weight is still the seeded orthogonal draw."""
    with torch.no_grad():
        model.actor_head[3].weight.mul_(120.0)
        model.critic_head[2].weight.mul_(6.0)
        model.critic_projector[0].weight.mul_(2.5)
        for j, (name, p) in enumerate(model.named_parameters()):
            if p.dim() == 1 and name.endswith("bias"):
                p.add_(0.1 * torch.sin(0.37 * torch.arange(p.numel(), dtype=torch.float64) + 1.3 * j).to(p.dtype))


def section_ppo_init(traj="traj_small_hunter.npz", out_name="ppo_init.npz", label="G7b", P=32):
    """G7b / G7c (the same on bloxCapture, 20 x 20: the board of the 28-tile tower kernels and the 416-token attention): the reference's MAPPOAgent with ITS OWN initialisation (orthogonal weights, pacman_mappo_resnet.py:149-158) under a
    fixed torch seed -- the seed and per-tensor checksums are stored, not the 2.6 M weights -- on 32 env-ticks x 2 learners of a
    real trajectory (paired rows 2k, 2k + 1 share merged input k), once as initialised ("init") and once after sharpen_init
    ("sharp": logits and values of O(1)).  Unlike the closed-form sine weights of G7, these are the weights training starts
    from, so a bf16 evaluation of this fixture is representative of the production path."""
    print(f"{label} PPO losses, reference initialisation ({traj})")
    with contextlib.redirect_stdout(io.StringIO()):
        import pacman_mappo_resnet as M
    seed = 20260
    torch.set_num_threads(1)
    z = np.load(os.path.join(OUT, traj))
    obs_all = z["obs"]  # [T,4,8,H,W] u8
    idx = np.arange(7, 7 + P * 9, 9)
    assert idx[-1] < obs_all.shape[0], (idx[-1], obs_all.shape)
    ob1 = torch.tensor(obs_all[idx, 1].astype(np.float32))
    ob3 = torch.tensor(obs_all[idx, 3].astype(np.float32))
    obs = torch.stack([ob1, ob3], dim=1).reshape((2 * P,) + tuple(ob1.shape[1:]))              # rows 2k, 2k+1 = the two blue learners
    merged = torch.stack([M.merge_obs_for_critic([ob1[i], ob3[i]]) for i in range(P)])           # [P]
    clip_eps, ent_coef, lr = M.CLIP_EPS, 0.02, 2e-4
    out = dict(obs=obs.numpy().astype(np.uint8), merged=merged.numpy().astype(np.uint8))
    names = None
    for tag in ("init", "sharp"):
        torch.manual_seed(seed)
        model = M.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
        if tag == "sharp":
            sharpen_init(model)
        names = [n for n, _ in model.named_parameters()]
        rng = np.random.RandomState(11)
        with torch.no_grad():
            logits0 = model.actor_head(model.actor_backbone(obs))
            lp0 = torch.log_softmax(logits0, -1)
        act = torch.tensor(rng.randint(5, size=2 * P), dtype=torch.long)
        old_logp = (lp0.gather(1, act.view(-1, 1)).squeeze(1) + torch.tensor((0.05 * rng.randn(2 * P)).astype(np.float32)))
        adv = torch.tensor(rng.randn(2 * P).astype(np.float32))
        ret = torch.tensor((0.5 * rng.randn(2 * P)).astype(np.float32))
        sums = [float(p.detach().double().sum()) for p in model.parameters()]
        abss = [float(p.detach().double().abs().sum()) for p in model.parameters()]
        vals, lps, ent = model.evaluate(obs, merged.repeat_interleave(2, dim=0), act)           # :556-557 expands the merged input per agent
        norm_adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        ratio = (lps - old_logp).exp()
        pg = -torch.min(norm_adv * ratio, norm_adv * torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps)).mean()
        vl = 0.5 * ((vals - ret) ** 2).mean()
        loss = pg + M.VF_COEF * vl - ent_coef * ent.mean()
        model.zero_grad()
        loss.backward()
        gnorms = [float(p.grad.double().norm()) for p in model.parameters()]
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), M.MAX_GRAD_NORM)
        rec = dict(act=act.numpy(), old_logp=old_logp.numpy(), adv=adv.numpy(), ret=ret.numpy(), logits=logits0.numpy(),
                   values=vals.detach().numpy(), logp=lps.detach().numpy(), entropy=ent.detach().numpy(),
                   pg=np.float32(pg.item()), vl=np.float32(vl.item()), loss=np.float32(loss.item()), grad_norm=np.float32(float(gn)),
                   param_sum=np.array(sums, np.float64), param_abs=np.array(abss, np.float64), grad_norms=np.array(gnorms, np.float64))
        out.update({f"{tag}_{k}": v for k, v in rec.items()})
        print(f"  {tag}: loss={loss.item():.6f} pg={pg.item():.6f} vl={vl.item():.6f} gn={float(gn):.6f} "
              f"|logits|max={float(logits0.abs().max()):.4f} values in [{float(vals.min()):.3f}, {float(vals.max()):.3f}] "
              f"entropy mean {float(ent.mean()):.4f}")
    out["meta"] = np.frombuffer(json.dumps(dict(
        seed=seed, clip_eps=clip_eps, ent_coef=ent_coef, lr=lr, vf_coef=M.VF_COEF, max_grad_norm=M.MAX_GRAD_NORM,
        weights="torch.manual_seed(seed); MAPPOAgent(obs_shape, 5, 2): the reference's orthogonal initialisation; 'sharp' = sharpen_init of it",
        param_names=names, torch_version=torch.__version__,
        source="pacman_mappo_resnet.MAPPOAgent.__init__/evaluate + PPO loss lines 577-590")).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, out_name), **out)
    print("  " + out_name)


def section_ppo_init_blox():
    section_ppo_init("traj_blox_hunter.npz", "ppo_init_blox.npz", "G7c")


def section_ppo_init_tiny():
    section_ppo_init("traj_tiny_hunter.npz", "ppo_init_tiny.npz", "G7d")


def section_shaping():
    print("G8 shaping / canonicalize / merge")
    with contextlib.redirect_stdout(io.StringIO()):
        import pacman_mappo_resnet as M
    z = np.load(os.path.join(OUT, "traj_small_hunter.npz"))
    obs = z["obs"]
    resets = z["resets"]
    T = 600
    shp = np.zeros((T, 4), np.float64)
    for t in range(T - 1):
        for i in range(4):
            red = i in (0, 2)
            cur = M.canonicalize_obs(torch.tensor(obs[t, i].astype(np.float32)), red)
            nxt = M.canonicalize_obs(torch.tensor(obs[t + 1, i].astype(np.float32)), red)
            shp[t, i] = M.compute_heuristic_shaping(cur, nxt)
    ticks = [0, 1, 5, 50, 333]
    canon = np.stack([np.stack([M.canonicalize_obs(torch.tensor(obs[t, i].astype(np.float32)), True).numpy().astype(np.uint8)
                                for i in range(4)]) for t in ticks])
    merged_blue = np.stack([M.merge_obs_for_critic([torch.tensor(obs[t, 1].astype(np.float32)),
                                                    torch.tensor(obs[t, 3].astype(np.float32))]).numpy().astype(np.uint8) for t in ticks])
    merged_red = np.stack([M.merge_obs_for_critic([M.canonicalize_obs(torch.tensor(obs[t, 0].astype(np.float32)), True),
                                                   M.canonicalize_obs(torch.tensor(obs[t, 2].astype(np.float32)), True)]).numpy().astype(np.uint8) for t in ticks])
    amap = np.array([M.canonicalize_action(a, True) for a in range(5)], np.int8)
    np.savez_compressed(os.path.join(OUT, "shaping.npz"), shaping=shp, ticks=np.array(ticks), canon_red=canon,
                        merged_blue=merged_blue, merged_red=merged_red, action_map_red=amap, T=np.int32(T),
                        meta=np.frombuffer(json.dumps(dict(
                            traj="traj_small_hunter.npz",
                            note="shaping[t,i] from consecutive obs[t,i], obs[t+1,i] of the trajectory (no reset handling, as recorded)",
                            source="pacman_mappo_resnet.compute_heuristic_shaping/canonicalize_obs/merge_obs_for_critic")).encode(), np.uint8))
    print("  shaping.npz")


def section_bots():
    print("G9 bot traces")
    for lay, team, seed, ticks in (("tiny", "baselineTeam", 0, 320), ("small", "baselineTeam", 1, 320),
                                   ("small", "randomTeam", 2, 320), ("tiny", "randomTeam", 3, 150)):
        random.seed(seed)
        env = make_env(lay, 299, self_play=False, enemy=team)
        rng = np.random.RandomState(1)
        quiet(env.reset)
        # hook the bots' getAction to record what they chose
        chosen = []
        for b in (env.agents[0], env.agents[2]):
            orig = b.getAction

            def wrap(gs, _o=orig):
                a = _o(gs)
                chosen.append(a)
                return a
            b.getAction = wrap
        trace, scores, rewards, dones = [], [], [], []
        blue = []
        for t in range(ticks):
            a1, a3 = int(rng.randint(5)), int(rng.randint(5))
            blue.append([a1, a3])
            chosen.clear()
            acts = {env.agents[1]: a1, env.agents[3]: a3}
            obs, rew, term, info = env.step(acts)
            trace.append([DIR2INT[c] for c in chosen])
            scores.append(int(env.game.state.data.score))
            rewards.append([float(rew[env.agents[0]]), float(rew[env.agents[1]])])
            d = bool(any(term.values()))
            dones.append(d)
            if d:
                quiet(env.reset)
                for b in (env.agents[0], env.agents[2]):
                    orig = b.getAction

                    def wrap(gs, _o=orig):
                        a = _o(gs)
                        chosen.append(a)
                        return a
                    b.getAction = wrap
        with open(os.path.join(OUT, f"bots_{lay}_{team}.json"), "w") as f:
            json.dump(dict(layout=layout_text(lay), team=team, random_seed=seed, length=299, blue_actions=blue,
                           red_actions=trace, scores=scores, rewards=rewards, dones=dones,
                           source="gymPacMan_parallel_env(self_play=False, enemieName=team); random.seed(k) before ctor; "
                                  "blue = numpy RandomState(1).randint(5) x2 per tick"), f)
        print(f"  bots_{lay}_{team}.json final scores {sorted(set(scores))[:6]} dones={sum(dones)}")


SECTIONS = dict(traj=section_traj, scen=section_scen, dist=section_dist, maze=section_maze, gae=section_gae,
                ppo=section_ppo, ppo_init=section_ppo_init, ppo_init_blox=section_ppo_init_blox, ppo_init_tiny=section_ppo_init_tiny, shaping=section_shaping, bots=section_bots)

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
