"""GPU parity tests proper: the HIP path (through the C ABI, via pmx.PmxVecEnv) against
  (a) the committed golden fixtures captured from the reference, and
  (b) the CPU oracle on identical seeded inputs,
bit-exact for every integer/byte/index output and for the float64 rewards.  Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest
import torch

import _golden as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _pmx():
    import pmx
    return pmx


def _state_tuple(s, H):
    return (tuple((s.pos[i][0], s.pos[i][1]) for i in range(4)), tuple(s.dir), tuple(s.pac), tuple(s.scared),
            tuple(s.carry), tuple(s.ret), tuple(s.food[:H]), tuple(s.caps[:H]), s.score, s.steps, s.ticks)


def _ostate_to_pmx(pmx, p, H):
    return pmx.make_state([(p.pos[i][0], p.pos[i][1]) for i in range(4)], p.dir, p.pac, p.scared, p.carry, p.ret,
                          p.food, p.caps, p.score, p.steps, H)


# ------------------------------------------------------------------------------------------------ golden replay
@pytest.mark.parametrize("name", G.names("traj_*.npz"))
def test_golden_trajectory(name):
    """Every env of a 96-env batch replays the recorded action stream; all outputs must equal the reference's."""
    pmx = _pmx()
    d, meta = G.load(name)
    N = 96
    lay = pmx.Layout.from_text(meta["layout"])
    env = pmx.PmxVecEnv(lay, N, length=meta["length"], reward_forLegalAction=meta["legal_reward"],
                        defenceReward=meta["defence"], auto_reset=False)
    H = lay.height
    obs, legal = env.reset()
    assert (obs.cpu().numpy() == d["init_obs"].astype(np.float32)[None]).all()
    assert (legal.cpu().numpy() == d["init_legal"][None]).all()
    T = len(d["actions"])
    for t in range(T):
        a = torch.tensor(d["actions"][t], dtype=torch.int8).repeat(N, 1).cuda()
        obs, rew, done, info = env.step(a)
        o = obs.cpu().numpy()
        assert (o == d["obs"][t].astype(np.float32)[None]).all(), f"{name} t={t} obs"
        r = rew.cpu().numpy()
        assert r.tobytes() == np.repeat(d["reward"][t][None], N, 0).tobytes(), f"{name} t={t} reward {r[0]} vs {d['reward'][t]}"
        assert (done.cpu().numpy() == d["done"][t]).all(), f"{name} t={t} done"
        assert (info["legal_actions"].cpu().numpy() == d["legal"][t][None]).all(), f"{name} t={t} legal"
        assert (info["score_change"].cpu().numpy() == d["score_change"][t]).all()
        assert (info["score"].cpu().numpy() == d["sub_score"][t, 3]).all()
        if t % 37 == 0 or d["resets"][t]:
            st = env.get_state(0, 2)
            for s in st:
                assert tuple((s.pos[i][0], s.pos[i][1]) for i in range(4)) == tuple(map(tuple, d["sub_pos"][t, 3]))
                assert tuple(s.food[:H]) == tuple(d["sub_food"][t, 3]) and tuple(s.caps[:H]) == tuple(d["sub_caps"][t, 3])
                assert tuple(s.carry) == tuple(d["sub_carry"][t, 3]) and tuple(s.ret) == tuple(d["sub_ret"][t, 3])
                assert tuple(s.scared) == tuple(d["sub_scared"][t, 3]) and tuple(s.dir) == tuple(d["sub_dir"][t, 3])
                assert tuple(s.pac) == tuple(d["sub_pac"][t, 3])
        if d["resets"][t]:
            env.reset()
    env.close()


@pytest.mark.parametrize("name", G.names("scen_*.npz"))
def test_golden_scenarios(name):
    """All K hand-built / randomised states of a fixture are loaded into a K-env batch and advanced one tick."""
    pmx = _pmx()
    d, meta = G.load(name)
    K = len(d["actions"])
    lay = pmx.Layout.from_text(meta["layout"])
    H = lay.height
    env = pmx.PmxVecEnv(lay, K, length=meta["length"], reward_forLegalAction=meta["legal_reward"],
                        defenceReward=meta["defence"], auto_reset=False)
    states = [pmx.make_state(d["in_pos"][k], d["in_dir"][k], d["in_pac"][k], d["in_scared"][k], d["in_carry"][k],
                             d["in_ret"][k], d["in_food"][k], d["in_caps"][k], d["in_score"][k], d["in_steps"][k], H)
              for k in range(K)]
    env.set_state(states)
    back = env.get_state()
    for k in range(0, K, 17):
        assert _state_tuple(back[k], H) == _state_tuple(states[k], H)
    obs, rew, done, info = env.step(torch.tensor(d["actions"], dtype=torch.int8).cuda())
    o = obs.cpu().numpy()
    bad = np.nonzero((o != d["obs"].astype(np.float32)).reshape(K, -1).any(1))[0]
    assert len(bad) == 0, f"{name}: obs differ for scenarios {bad[:10]} {[meta['names'][b] for b in bad[:5]]}"
    r = rew.cpu().numpy()
    badr = np.nonzero((r.view(np.uint64) != d["reward"].view(np.uint64)).any(1))[0]
    assert len(badr) == 0, f"{name}: rewards differ for {badr[:10]}: {r[badr[:3]]} vs {d['reward'][badr[:3]]}"
    assert (done.cpu().numpy() == d["done"]).all()
    assert (info["legal_actions"].cpu().numpy() == d["legal"]).all()
    assert (info["score_change"].cpu().numpy() == d["score_change"]).all()
    st = env.get_state()
    for k in range(K):
        s = st[k]
        tag = f"{name} k={k} {meta['names'][k]}"
        assert tuple((s.pos[i][0], s.pos[i][1]) for i in range(4)) == tuple(map(tuple, d["sub_pos"][k, 3])), tag
        assert tuple(s.food[:H]) == tuple(d["sub_food"][k, 3]), tag
        assert tuple(s.caps[:H]) == tuple(d["sub_caps"][k, 3]), tag
        assert tuple(s.carry) == tuple(d["sub_carry"][k, 3]) and tuple(s.ret) == tuple(d["sub_ret"][k, 3]), tag
        assert tuple(s.scared) == tuple(d["sub_scared"][k, 3]) and tuple(s.pac) == tuple(d["sub_pac"][k, 3]), tag
        assert tuple(s.dir) == tuple(d["sub_dir"][k, 3]) and s.score == d["sub_score"][k, 3], tag
        assert s.steps == d["in_steps"][k] + 1, tag
    env.close()


# ------------------------------------------------------------------------------------- differential vs the oracle
def _seed_states(pmx, env, orc, fixture, H):
    """Start the batch from the fixture's randomised states (cycled) so that collisions/returns happen early."""
    d, _ = G.load(fixture)
    K = len(d["actions"])
    N = env.n_envs
    states = []
    for e in range(N):
        k = e % K
        if e % 5 == 4:
            states.append(None)  # keep the initial state
            continue
        states.append(pmx.make_state(d["in_pos"][k], d["in_dir"][k], d["in_pac"][k], d["in_scared"][k], d["in_carry"][k],
                                     d["in_ret"][k], d["in_food"][k], d["in_caps"][k], d["in_score"][k],
                                     min(int(d["in_steps"][k]), 250), H))
    cur = env.get_state()
    full = [states[e] if states[e] is not None else cur[e] for e in range(N)]
    env.set_state(full)
    for e in range(N):
        s = full[e]
        p = O.PState()
        for i in range(4):
            p.pos[i][0], p.pos[i][1] = s.pos[i][0], s.pos[i][1]
            p.dir[i], p.pac[i], p.scared[i], p.carry[i], p.ret[i] = s.dir[i], s.pac[i], s.scared[i], s.carry[i], s.ret[i]
        for y in range(H):
            p.food[y], p.caps[y] = s.food[y], s.caps[y]
        p.score, p.steps = s.score, s.steps
        orc.set_state(e, p)


@pytest.mark.parametrize("layname,fixture,N,T,dtype", [
    ("small", "scen_small_random.npz", 2048, 330, "float32"),
    ("tiny", "scen_tiny_random.npz", 1024, 200, "uint8"),
    ("maze23", "scen_maze23_random.npz", 1024, 200, "bfloat16"),
    ("blox", "scen_blox_random.npz", 512, 120, "float32"),
])
def test_differential_vs_oracle(layname, fixture, N, T, dtype):
    pmx = _pmx()
    _, meta = G.load(fixture)
    rows = meta["layout"]
    lay = pmx.Layout.from_text(rows)
    H, W = lay.height, lay.width
    length = 300
    env = pmx.PmxVecEnv(lay, N, length=length, auto_reset=True, obs_dtype=dtype, seed=77)
    orc = O.BatchEnv(rows, N, length=length, auto_reset=True, seed=77)
    env.reset()
    _seed_states(pmx, env, orc, fixture, H)
    rng = np.random.RandomState(5)
    oobs = np.zeros((N, 4, 8, H, W), np.float32)
    legal = np.stack([[orc.lib.orc_legal(orc.L.buf, C.byref(orc.S, e * orc.ssz), i) for i in range(4)] for e in range(N)]).astype(np.uint8)
    n_done = 0
    for t in range(T):
        # 70 % legal-uniform, 30 % anything (incl. illegal and out-of-range codes)
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        pick = rng.rand(N, 4) < 0.7
        for _ in range(3):
            ill = pick & (((legal >> np.clip(a, 0, 4)) & 1) == 0)
            a[ill] = rng.randint(0, 5, size=int(ill.sum()))
        a[rng.rand(N, 4) < 0.01] = rng.choice([-1, 5, 7, 127, -128])
        a[rng.rand(N, 4) < 0.12] = -2          # PMX_ACTION_RANDOM_LEGAL: the in-kernel randomTeam opponent
        if t % 40 == 7:
            a[:, 0] = -2; a[:, 2] = -2         # whole red team random, as in the curriculum's first phase
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        torch.cuda.synchronize()
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), f"t={t} reward"
        assert (done.cpu().numpy() == orc.done).all(), f"t={t} done"
        assert (info["legal_actions"].cpu().numpy() == orc.legal).all(), f"t={t} legal"
        assert (info["score_change"].cpu().numpy() == orc.score_change).all(), f"t={t} score_change"
        assert (info["score"].cpu().numpy() == orc.score).all(), f"t={t} score"
        assert (info["agent"].cpu().numpy().astype(np.uint32) == orc.agent).all(), f"t={t} agent words"
        o = obs.float().cpu().numpy()
        bad = np.nonzero((o != oobs).reshape(N, -1).any(1))[0]
        assert len(bad) == 0, f"t={t}: obs differ for envs {bad[:10]}"
        legal = orc.legal.copy()
        n_done += int(orc.done.sum())
        if t % 50 == 49:
            st = env.get_state()
            for e in range(0, N, 7):
                assert _state_tuple(st[e], H) == _state_tuple(orc.get_state(e), H), f"t={t} env={e}"
    assert n_done > 0
    env.close()


# ------------------------------------------------------------------------------------------ API-level behaviour
def test_step_agent_equals_step():
    pmx = _pmx()
    _, meta = G.load("scen_small_random.npz")
    lay = pmx.Layout.from_text(meta["layout"])
    N, H = 512, lay.height
    a_env = pmx.PmxVecEnv(lay, N, length=40, auto_reset=False)
    b_env = pmx.PmxVecEnv(lay, N, length=40, auto_reset=False)
    orc = O.BatchEnv(meta["layout"], N, length=40, auto_reset=False)
    _seed_states(pmx, a_env, orc, "scen_small_random.npz", H)
    b_env.set_state(a_env.get_state())
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(45):
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        obs, rew, done, info = a_env.step(a)
        sub = [b_env.step_agent(i, a[:, i].contiguous()).clone() for i in range(4)]
        for i in range(4):
            assert torch.equal(sub[i], obs[:, i]), (t, i)
        assert torch.equal(b_env.reward, rew) and torch.equal(b_env.done, done)
        assert torch.equal(b_env.legal, info["legal_actions"]) and torch.equal(b_env.score_change, info["score_change"])
        assert torch.equal(b_env.agent, info["agent"])
    with pytest.raises(pmx.PmxError):
        b_env.step_agent(2, a[:, 2].contiguous())   # sub-steps must come in order 0,1,2,3
    a_env.close(); b_env.close()


def test_reset_mask_observe_and_subset():
    pmx = _pmx()
    lay = pmx.get_layout("tinyCapture")
    N = 300   # not a multiple of 64/128/256: tail handling
    env = pmx.PmxVecEnv(lay, N, length=20, auto_reset=False)
    sub = pmx.PmxVecEnv(lay, N, length=20, auto_reset=False, obs_agents=(1, 3), obs_dtype="uint8")
    g = torch.Generator(device="cuda").manual_seed(9)
    init_obs = env.reset()[0].clone()
    for t in range(10):
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        obs, *_ = env.step(a)
        sobs, *_ = sub.step(a)
        assert sobs.shape == (N, 2, 8, lay.height, lay.width)
        assert torch.equal(sobs.float(), obs[:, [1, 3]])
    stepped = env.observe()[0].clone()
    # observe() = encode the final state for all agents: agent 3's planes of the last step saw that state
    assert torch.equal(stepped[:, 3], obs[:, 3])
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    o2, _ = env.reset(mask)
    assert torch.equal(o2[::3], init_obs[::3])
    keep = torch.ones(N, dtype=torch.bool); keep[::3] = False
    assert torch.equal(o2[keep], stepped[keep])
    env.close(); sub.close()


def test_auto_reset_returns_fresh_observations():
    pmx = _pmx()
    lay = pmx.get_layout("smallCapture")
    N = 256
    env = pmx.PmxVecEnv(lay, N, length=5, auto_reset=True)
    init = env.reset()[0][0].clone()
    a = torch.full((N, 4), 4, dtype=torch.int8, device="cuda")
    for t in range(6):
        obs, rew, done, info = env.step(a)
        assert int(done.sum()) == (N if t == 5 else 0)     # length + 1 ticks (gymPacMan.py:268 before :189)
    assert torch.equal(obs, init[None].expand_as(obs))
    st = env.get_state(0, 3)
    assert all(s.steps == 0 and s.score == 0 for s in st)
    env.close()


def test_invalid_layouts_raise():
    pmx = _pmx()
    rows = ["%%%%%%%%%%", "%1 .  . 2%", "%3      4%", "%%%%%%%%%%"]
    pmx.PmxVecEnv(pmx.Layout.from_text(rows), 4).close()
    bad_border = ["%%%%%%%%%%", "%1 .  . 2.", "%3      4%", "%%%%%%%%%%"]
    with pytest.raises(pmx.PmxError):
        pmx.PmxVecEnv(pmx.Layout.from_text(bad_border), 4)
    swapped = ["%%%%%%%%%%", "%2 .  . 1%", "%3      4%", "%%%%%%%%%%"]
    with pytest.raises(pmx.PmxError):
        pmx.PmxVecEnv(pmx.Layout.from_text(swapped), 4)


# --------------------------------------------------------------------------------- full-size, size-independent
@pytest.mark.parametrize("layname,N", [("smallCapture", 16384), ("tinyCapture", 4096), ("smallCapture", 8192)])
def test_full_size_properties(layname, N):
    """BASELINE.json sizes (configs 3 and 2, and the 8 192-env per-GPU shard of config 4).  Properties that need no reference at
    that size:
       replication (the same action stream in every 64-env tile gives identical results), pellet conservation
       (food on board + carried + returned == layout total), and observation/state consistency."""
    pmx = _pmx()
    lay = pmx.get_layout(layname)
    H, W = lay.height, lay.width
    env = pmx.PmxVecEnv(lay, N, length=300, auto_reset=True)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(11)
    tile = 64
    walls = torch.tensor([[(int(lay.wall_rows[y]) >> x) & 1 for x in range(W)] for y in range(H)], dtype=torch.float32).cuda()
    for t in range(330):
        a_tile = torch.randint(0, 5, (tile, 4), generator=g, device="cuda", dtype=torch.int8)
        a = a_tile.repeat(N // tile, 1)
        obs, rew, done, info = env.step(a)
        v = obs.view(N // tile, tile, 4, 8, H, W)
        assert torch.equal(v, v[:1].expand_as(v)), f"t={t}: tiles diverged"
        assert torch.equal(rew.view(N // tile, tile, 2), rew[:tile][None].expand(N // tile, tile, 2))
        if t % 25 == 0:
            assert torch.equal(obs[:, :, 0], walls[None, None].expand(N, 4, H, W))
            self_plane = obs[:, :, 1]
            assert torch.equal((self_plane > 0).sum((-1, -2)), torch.ones(N, 4, device="cuda", dtype=torch.long))
            food_cells = (obs[:, 3, 6] + obs[:, 3, 7]).sum((-1, -2))
            carry = torch.stack([obs[:, i, 1].amax((-1, -2)) - 1 for i in range(4)], 1)
            st = env.get_state(0, 64)
            for e in range(64):
                s = st[e]
                onboard = sum(bin(s.food[y]).count("1") for y in range(H))
                assert onboard == int(food_cells[e].item())
                assert onboard + sum(s.carry) + sum(s.ret) == lay.total_food, f"t={t} env={e}: pellets not conserved"
                assert int(carry[e, 3].item()) == s.carry[3]
    env.close()


def test_full_size_properties_config5_shard():
    """BASELINE config 5's per-GPU shard: 8 192 envs, each on its OWN generated 20 x 20 maze (mazeGenerator seeds 1..8192).
    Size-independent properties: every env shows the walls of its own maze, exactly one self cell per agent, pellets are
    conserved against its own maze's total (food on board + carried + returned), carried counts agree between planes and
    state, and a second handle fed the same action stream produces identical planes and rewards (no cross-env interference)."""
    pmx = _pmx()
    from pmx import maze_generator
    N = 8192
    lays = [pmx.Layout.from_text(maze_generator.generate_maze(seed)) for seed in range(1, N + 1)]
    H, W = lays[0].height, lays[0].width
    assert (H, W) == (20, 20)
    envs = [pmx.PmxVecEnv(lays, N, length=60, auto_reset=True, obs_dtype="uint8", seed=5) for _ in range(2)]
    for e in envs:
        e.reset()
    walls = torch.tensor(np.stack([[[(int(l.wall_rows[y]) >> x) & 1 for x in range(W)] for y in range(H)] for l in lays]),
                         dtype=torch.uint8).cuda()
    totals = [l.total_food for l in lays]
    g = torch.Generator(device="cuda").manual_seed(17)
    for t in range(150):                                    # two and a half episodes: resets happen at full size too
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        o0, r0, d0, i0 = envs[0].step(a)
        o1, r1, d1, i1 = envs[1].step(a)
        assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1), f"t={t}: the two handles diverged"
        if t % 30 == 0 or t == 149:
            assert torch.equal(o0[:, :, 0], walls[:, None].expand(N, 4, H, W))
            assert torch.equal((o0[:, :, 1] > 0).sum((-1, -2)), torch.ones(N, 4, device="cuda", dtype=torch.long))
            food_cells = (o0[:, 3, 6].long() + o0[:, 3, 7].long()).sum((-1, -2))
            carry3 = o0[:, 3, 1].amax((-1, -2)).long() - 1
            for first in (0, N - 96):
                st = envs[0].get_state(first, 96)
                for k in range(96):
                    s, e = st[k], first + k
                    onboard = sum(bin(s.food[y]).count("1") for y in range(H))
                    assert onboard == int(food_cells[e].item())
                    assert onboard + sum(s.carry) + sum(s.ret) == totals[e], f"t={t} env={e}: pellets not conserved"
                    assert int(carry3[e].item()) == s.carry[3]
    for e in envs:
        e.close()


# ------------------------------------------------------------------------------------------- other kernels
@pytest.mark.parametrize("lay", ["tiny", "small", "blox", "maze23"])
def test_maze_distances_golden(lay):
    pmx = _pmx()
    d, meta = G.load(f"dist_{lay}.npz")
    env = pmx.PmxVecEnv(pmx.Layout.from_text(meta["layout"]), 1)
    cells, dist = env.maze_distances()
    assert (cells.cpu().numpy() == d["cells"]).all()
    assert (dist.cpu().numpy() == d["dist"]).all()
    env.close()


def _gae(mode, rew, val, done, last, gamma, lam):
    pmx = _pmx()
    lib = pmx._lib.load()
    T, n = rew.shape
    adv = torch.empty_like(rew); ret = torch.empty_like(rew)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if mode is None:
        rc = lib.pmx_gae(rew.data_ptr(), val.data_ptr(), done.data_ptr(), last.data_ptr(), T, n, gamma, lam, adv.data_ptr(), ret.data_ptr(), st)
    else:
        rc = lib.pmx_gae_mode(rew.data_ptr(), val.data_ptr(), done.data_ptr(), last.data_ptr(), T, n, gamma, lam, adv.data_ptr(), ret.data_ptr(), mode, st)
    assert rc == 0
    return adv, ret


def test_gae_golden_bit_exact_and_wave_scan():
    d, meta = G.load("gae.npz")
    for c in meta["cases"]:
        rew = torch.tensor(d[c + "_rew"]).cuda()[:, None].contiguous()
        val = torch.tensor(d[c + "_val"]).cuda()[:, None].contiguous()
        done = torch.tensor(d[c + "_done"]).cuda()[:, None].contiguous()
        last = torch.tensor([float(d[c + "_last"])], dtype=torch.float32).cuda()
        adv, ret = _gae(0, rew, val, done, last, meta["gamma"], meta["lam"])
        assert adv[:, 0].cpu().numpy().tobytes() == d[c + "_adv"].tobytes(), c      # lane kernel: bit-exact
        assert ret[:, 0].cpu().numpy().tobytes() == d[c + "_ret"].tobytes(), c
        adv2, ret2 = _gae(1, rew, val, done, last, meta["gamma"], meta["lam"])      # wave scan: re-associated floats
        scale = float(np.abs(d[c + "_adv"]).max()) + 1.0
        assert float((adv2[:, 0].cpu() - torch.tensor(d[c + "_adv"])).abs().max()) <= 1e-5 * scale, c
        assert float((ret2[:, 0].cpu() - torch.tensor(d[c + "_ret"])).abs().max()) <= 1e-5 * scale, c


def test_gae_batched_vs_oracle():
    rng = np.random.RandomState(0)
    T, n = 32, 4096
    rew = rng.randn(T, n).astype(np.float32); val = rng.randn(T, n).astype(np.float32)
    done = (rng.rand(T, n) < 0.05).astype(np.float32); last = rng.randn(n).astype(np.float32)
    adv, ret = _gae(None, torch.tensor(rew).cuda(), torch.tensor(val).cuda(), torch.tensor(done).cuda(), torch.tensor(last).cuda(), 0.99, 0.95)
    adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
    for i in range(0, n, 97):
        a, r = O.gae(rew[:, i], val[:, i], done[:, i], float(last[i]), 0.99, 0.95)
        assert a.tobytes() == adv[:, i].tobytes() and r.tobytes() == ret[:, i].tobytes()


def test_canonicalize_merge_golden():
    pmx = _pmx()
    lib = pmx._lib.load()
    d, meta = G.load("shaping.npz")
    tr, _ = G.load(meta["traj"])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for code, dt in ((0, torch.float32), (1, torch.bfloat16), (2, torch.uint8)):
        for j, t in enumerate(d["ticks"]):
            o = torch.tensor(tr["obs"][t]).cuda().to(dt).contiguous()      # [4,8,H,W]
            H, W = o.shape[-2:]
            canon = torch.empty_like(o)
            assert lib.pmx_canonicalize_obs(o.data_ptr(), canon.data_ptr(), 4, H, W, code, st) == 0
            assert (canon.float().cpu().numpy() == d["canon_red"][j]).all()
            mb = torch.empty_like(o[0])
            assert lib.pmx_merge_obs(o[1].data_ptr(), o[3].data_ptr(), mb.data_ptr(), 1, H, W, code, st) == 0
            assert (mb.float().cpu().numpy() == d["merged_blue"][j]).all()
            mr = torch.empty_like(o[0])
            assert lib.pmx_merge_obs(canon[0].data_ptr(), canon[2].data_ptr(), mr.data_ptr(), 1, H, W, code, st) == 0
            assert (mr.float().cpu().numpy() == d["merged_red"][j]).all()


def test_per_env_layouts_vs_oracle():
    """BASELINE config 5 in miniature: every env plays its own generated maze (20x20, 2 capsules each)."""
    pmx = _pmx()
    from pmx import maze_generator as MG
    n_lay, N, T = 48, 768, 260
    rows = [MG.generate_maze(seed).split("\n") for seed in range(1, n_lay + 1)]
    lays = [pmx.Layout.from_text(r) for r in rows]
    rng = np.random.RandomState(9)
    index = rng.randint(0, n_lay, size=N).astype(np.int32)
    env = pmx.PmxVecEnv(lays, N, length=120, auto_reset=True, obs_dtype="bfloat16", seed=5, layout_index=index)
    orc = O.MultiBatchEnv(rows, index, length=120, auto_reset=True, seed=5)
    obs0, legal0 = env.reset()
    oobs = np.zeros((N, 4, 8, 20, 20), np.float32)
    # initial observations: plane 0 must be each env's own walls
    for e in range(0, N, 97):
        lay = lays[index[e]]
        w = np.array([[lay.is_wall(x, y) for x in range(20)] for y in range(20)], np.float32)
        assert (obs0[e, 0, 0].float().cpu().numpy() == w).all()
    for t in range(T):
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        a[rng.rand(N, 4) < 0.5] = -2                     # random-legal moves keep the agents wandering
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all() and (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        assert (info["agent"].cpu().numpy().astype(np.uint32) == orc.agent).all(), t
        bad = np.nonzero((obs.float().cpu().numpy() != oobs).reshape(N, -1).any(1))[0]
        assert len(bad) == 0, f"t={t}: obs differ for envs {bad[:8]} (layouts {index[bad[:8]]})"
    cells, dist = env.maze_distances(layout=7)
    oc, od = O.maze_distances(rows[7])
    assert (cells.cpu().numpy() == oc).all() and (dist.cpu().numpy() == od).all()
    env.close()


def test_one_distinct_maze_per_env_vs_oracle():
    """BASELINE config 5's shape: as many distinct generated mazes as envs (seeds 1..1024), one env each."""
    pmx = _pmx()
    from pmx import maze_generator as MG
    N, T = 1024, 150
    rows = [MG.generate_maze(seed).split("\n") for seed in range(1, N + 1)]
    lays = [pmx.Layout.from_text(r) for r in rows]
    index = np.arange(N, dtype=np.int32)
    env = pmx.PmxVecEnv(lays, N, length=60, auto_reset=True, obs_dtype="uint8", seed=11, layout_index=index)
    orc = O.MultiBatchEnv(rows, index, length=60, auto_reset=True, seed=11)
    env.reset()
    oobs = np.zeros((N, 4, 8, 20, 20), np.float32)
    rng = np.random.RandomState(4)
    for t in range(T):
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        a[rng.rand(N, 4) < 0.6] = -2
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all() and (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        bad = np.nonzero((obs.cpu().numpy() != oobs.astype(np.uint8)).reshape(N, -1).any(1))[0]
        assert len(bad) == 0, f"t={t}: obs differ for envs {bad[:8]}"
    env.close()


def _redraw_index(seed, env, ticks, n):
    """include/pmx.h redraw_layouts: the counter-based draw, restated in Python."""
    M = 0xFFFFFFFF
    x = ((seed ^ ((env * 0x9E3779B1) & M)) ^ ((ticks * 0x85EBCA77) & M) ^ 0x4C41594F) & M
    x ^= x >> 16; x = (x * 0x7FEB352D) & M; x ^= x >> 15; x = (x * 0x846CA68B) & M; x ^= x >> 16
    return (x * n) >> 32


def test_layout_redrawn_at_every_reset_vs_oracle():
    """random_layout=True (gymPacMan.py:98-100: a new maze at every reset()): a pool of 40 generated mazes, 640 envs, short
    episodes with in-kernel bots on both sides (thousands of auto-resets, each moving the env to a freshly drawn maze of the
    pool) -- every output of every tick against the oracle, which draws with the same counter-based generator; then an
    explicit masked pmx_reset against the generator restated in Python."""
    pmx = _pmx()
    from pmx import maze_generator as MG
    P, N, T, seed = 40, 640, 130, 21
    rows = [MG.generate_maze(100 + k).split("\n") for k in range(P)]
    lays = [pmx.Layout.from_text(r) for r in rows]
    index = (np.arange(N) % P).astype(np.int32)
    env = pmx.PmxVecEnv(lays, N, length=14, auto_reset=True, obs_dtype="uint8", seed=seed, layout_index=index, redraw_layouts=True)
    orc = O.MultiBatchEnv(rows, index.copy(), length=14, auto_reset=True, seed=seed, redraw=True)
    env.reset()
    assert not (env.layout_indices() == index).all()             # reset() itself draws (gymPacMan.reset does)
    exp0 = np.array([_redraw_index(seed, e, 0, P) for e in range(N)], np.int32)
    assert (env.layout_indices() == exp0).all()
    # bring the oracle to the same starting point: its constructor reset does not redraw, so rebuild it on the drawn layouts
    orc = O.MultiBatchEnv(rows, exp0.copy(), length=14, auto_reset=True, seed=seed, redraw=True)
    oobs = np.zeros((N, 4, 8, 20, 20), np.float32)
    rng = np.random.RandomState(8)
    moved = 0
    for t in range(T):
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        a[rng.rand(N, 4) < 0.6] = -2
        before = orc.index.copy()
        orc.tick(a, oobs)
        moved += int((before != orc.index).sum())
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all() and (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        bad = np.nonzero((obs.cpu().numpy() != oobs.astype(np.uint8)).reshape(N, -1).any(1))[0]
        assert len(bad) == 0, f"t={t}: obs differ for envs {bad[:8]}"
        if t % 15 == 14:
            assert (env.layout_indices() == orc.index).all(), t
    assert moved > 4 * N and len(set(orc.index.tolist())) > P // 2   # each env changed maze several times, the pool is used
    # explicit reset of a subset: those envs draw with their current tick counter, the others keep their maze
    cur = env.layout_indices()
    ticks = np.array([s.ticks for s in env.get_state(0, N)], np.uint32)
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    env.reset(mask)
    new = env.layout_indices()
    for e in range(N):
        want = _redraw_index(seed, e, int(ticks[e]), P) if e % 3 == 0 else cur[e]
        assert new[e] == want, e
    with pytest.raises(ValueError):
        pmx.PmxVecEnv(lays[0], 8, redraw_layouts=True)
    env.close()


_TINY_BOARD = ["%%%%%%%%", "%1 .. 2%", "%  ..  %", "%3 .. 4%", "%%%%%%%%"]


@pytest.mark.parametrize("rows_cols,dtype,N", [((14, 15), "uint8", 333), (None, "float32", 130), ((30, 15), "bfloat16", 257)])
def test_other_board_sizes_vs_oracle(rows_cols, dtype, N):
    """Boards at the edges of the supported range (32 wide: every shift count of the bit rows is exercised; 8 x 5;
    32 x 32) with batch sizes that are not multiples of the wave / block size."""
    pmx = _pmx()
    from pmx import maze_generator as MG
    rows = _TINY_BOARD if rows_cols is None else MG.generate_maze(11, rows=rows_cols[0], cols=rows_cols[1]).split("\n")
    lay = pmx.Layout.from_text(rows)
    H, W = lay.height, lay.width
    env = pmx.PmxVecEnv(lay, N, length=60, auto_reset=True, obs_dtype=dtype, seed=1)
    orc = O.BatchEnv(rows, N, length=60, auto_reset=True, seed=1)
    env.reset()
    rng = np.random.RandomState(2)
    oobs = np.zeros((N, 4, 8, H, W), np.float32)
    for t in range(150):
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        a[rng.rand(N, 4) < 0.6] = -2
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all() and (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        assert (obs.float().cpu().numpy() == oobs).all(), t
    cells, dist = env.maze_distances()
    oc, od = O.maze_distances(rows)
    assert (cells.cpu().numpy() == oc).all() and (dist.cpu().numpy() == od).all()
    env.close()


@pytest.mark.parametrize("layname", ["smallCapture", "tinyCapture"])
def test_in_kernel_baseline_bots_vs_oracle(layname):
    """Action codes -3 / -4 (the reference's reflex agents evaluated in the kernel on the mid-tick state) against the oracle's
    restatement, which tests/test_oracle_golden.py pins to the reference's own bot traces (fixture G9)."""
    pmx = _pmx()
    lay = pmx.get_layout(layname)
    N, T = 320, 340
    env = pmx.PmxVecEnv(lay, N, length=150, auto_reset=True, seed=21, bots=True)
    orc = O.BatchEnv(lay.text, N, length=150, auto_reset=True, seed=21)
    O.set_bot_tables(O.bot_tables(lay.text))
    env.reset()
    rng = np.random.RandomState(4)
    oobs = np.zeros((N, 4, 8, lay.height, lay.width), np.float32)
    score_seen = set()
    for t in range(T):
        a = rng.randint(0, 5, size=(N, 4)).astype(np.int8)
        a[:, 0] = -3                                   # red: baselineTeam (offensive agent 0, defensive agent 2)
        a[:, 2] = -4
        a[: N // 2, 1] = -3                            # half of the envs: bots on both sides, blue 1 offensive, blue 3 defensive
        a[: N // 2, 3] = -4
        a[N // 2:, 1][rng.rand(N - N // 2) < 0.5] = -2
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all() and (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        assert (info["agent"].cpu().numpy().astype(np.uint32) == orc.agent).all(), t
        assert (info["score"].cpu().numpy() == orc.score).all(), t
        assert (obs.cpu().numpy() == oobs).all(), t
        score_seen.update(orc.score.tolist())
    assert len(score_seen) >= 2                        # the bots do eat and return food (the reference scores 11 at once on tiny)
    env.close()


@pytest.mark.parametrize("layname", ["smallCapture", "bloxCapture"])
def test_soak_four_episodes_mixed_controllers_vs_oracle(layname):
    """A long run (4 full episodes with auto-reset, 1 210 ticks) in which every env mixes the three kinds of controllers that
    produce the rare rule paths -- reflex bots that chase, kill and carry food home, random-legal wanderers and raw (also
    illegal / out-of-range) action codes -- compared with the oracle on every output of every tick (state compared at the end)."""
    pmx = _pmx()
    lay = pmx.get_layout(layname)
    N, T = 192, 1210
    env = pmx.PmxVecEnv(lay, N, length=300, auto_reset=True, seed=77, bots=True, obs_dtype="uint8")
    orc = O.BatchEnv(lay.text, N, length=300, auto_reset=True, seed=77)
    O.set_bot_tables(O.bot_tables(lay.text))
    env.reset()
    rng = np.random.RandomState(123)
    kind = rng.randint(0, 4, size=(N, 4))              # per (env, agent): 0 offensive bot, 1 defensive bot, 2 random-legal, 3 raw codes
    oobs = np.zeros((N, 4, 8, lay.height, lay.width), np.float32)
    n_done = 0
    deaths = 0
    prev_carry = np.zeros((N, 4), np.int64)
    for t in range(T):
        raw = rng.randint(-1, 7, size=(N, 4)).astype(np.int8)
        a = np.where(kind == 0, -3, np.where(kind == 1, -4, np.where(kind == 2, -2, raw))).astype(np.int8)
        orc.tick(a, oobs)
        obs, rew, done, info = env.step(torch.tensor(a).cuda())
        assert rew.cpu().numpy().tobytes() == orc.reward.tobytes(), t
        assert (done.cpu().numpy() == orc.done).all(), t
        assert (info["legal_actions"].cpu().numpy() == orc.legal).all(), t
        assert (info["agent"].cpu().numpy().astype(np.uint32) == orc.agent).all(), t
        assert (info["score"].cpu().numpy() == orc.score).all() and (info["score_change"].cpu().numpy() == orc.score_change).all(), t
        if t % 7 == 0 or orc.done.any():
            assert (obs.cpu().numpy() == oobs.astype(np.uint8)).all(), t
        carry = (orc.agent >> 16).astype(np.int64)
        deaths += int(((prev_carry >= 2) & (carry == 0) & (orc.score_change[:, None] == 0)).sum())
        prev_carry = carry
        n_done += int(orc.done.sum())
    assert n_done >= 4 * N                              # at least the four time-outs; the bots also end games early by winning
    assert deaths > 0                                   # carriers did lose their food (kills with dumps happened)
    env.close()


def _emit_check(env, team, merged, raw, tag):
    """pmx_emit_team_obs of the env's current snapshots == canonicalize_obs / merge_obs_for_critic of the planes `raw`, both colours"""
    from pmx import trainer
    for red in (False, True):
        ids = [0, 2] if red else [1, 3]
        want = raw[:, ids].float()
        if red:
            want = trainer.canonicalize_obs(want)
        want_m = trainer.merge_obs(want[:, 0].contiguous(), want[:, 1].contiguous())
        team.fill_(9); merged.fill_(9)
        env.emit_team_obs(red, team, merged)
        assert torch.equal(team.float(), want), (tag, red)
        assert torch.equal(merged.float(), want_m), (tag, red)


@pytest.mark.parametrize("case", ["ragged131", "remapped256", "mazes256"])
def test_emit_team_obs_env_counts_and_per_env_layouts(case):
    """The one-wave-per-env kernel's block order depends on the env count (multiples of 128 are dealt to the XCDs in groups of
    the 16 envs that share a line of snapshot words; other counts are not, and a count that is no multiple of 4 leaves waves of the
    last block idle), and with per-env layouts every wave reads its own layout record: byte planes, both colours, 40 ticks."""
    import pmx
    if case == "mazes256":
        from pmx import maze_generator
        N = 256
        lay = [pmx.Layout.from_text(maze_generator.generate_maze(seed)) for seed in range(1, N + 1)]
    else:
        N = 131 if case == "ragged131" else 256
        lay = "smallCapture"
    env = pmx.PmxVecEnv(lay, N, length=25, auto_reset=True, obs_dtype="uint8", seed=3)
    H, W = env.layout.height, env.layout.width
    team = torch.empty((N, 2, 8, H, W), dtype=torch.uint8, device="cuda")
    merged = torch.empty((N, 8, H, W), dtype=torch.uint8, device="cuda")
    guard = torch.full((4096,), 9, dtype=torch.uint8, device="cuda")     # allocated right behind: a store past the end would land here
    obs, _ = env.reset()
    _emit_check(env, team, merged, obs.clone(), "reset")
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(40):
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        obs, _, _, _ = env.step(a)
        _emit_check(env, team, merged, obs.clone(), t)
    assert int(guard.min()) == 9 and int(guard.max()) == 9
    env.close()


@pytest.mark.parametrize("layout", ["smallCapture", "tinyCapture", "bloxCapture"])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "uint8"])
def test_emit_team_obs_equals_canonicalize_and_merge_of_the_tick_observations(layout, dtype):
    """pmx_emit_team_obs against the step's own four observations pushed through canonicalize_obs / merge_obs_for_critic (the
    torch forms are pinned by fixture G8 in tests/test_gpu_trainer.py): both colours, every tick of 70 with random play and
    auto-reset (staggered snapshots, carried food, resets), right after a reset, and with both learners on ONE cell carrying
    different amounts (plane 1 of the merged input is a max)."""
    import pmx
    from pmx import trainer
    N = 192
    env = pmx.PmxVecEnv(layout, N, length=25, auto_reset=True, obs_dtype=dtype, seed=9)
    H, W = env.layout.height, env.layout.width
    team = torch.empty((N, 2, 8, H, W), dtype=env.obs_torch_dtype, device="cuda")
    merged = torch.empty((N, 8, H, W), dtype=env.obs_torch_dtype, device="cuda")

    def check(raw, tag):
        for red in (False, True):
            ids = [0, 2] if red else [1, 3]
            want = raw[:, ids].float()
            if red:
                want = trainer.canonicalize_obs(want)
            want_m = trainer.merge_obs(want[:, 0].contiguous(), want[:, 1].contiguous())
            env.emit_team_obs(red, team, merged)
            assert torch.equal(team.float(), want), (tag, red)
            assert torch.equal(merged.float(), want_m), (tag, red)
            team.fill_(7); env.emit_team_obs(red, team, None)
            assert torch.equal(team.float(), want), (tag, red, "no merged")

    obs, _ = env.reset()
    check(obs.clone(), "reset")
    g = torch.Generator(device="cuda").manual_seed(4)
    for t in range(70):
        # mostly legal-ish random play with a bias to the east/west so that agents cross the border and carry food
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        obs, _, _, _ = env.step(a)
        check(obs.clone(), t)
    # both blue learners (and both red ones) on one cell with different loads
    st = env.get_state(0, N)
    for k in range(N):
        for i, j in ((1, 3), (0, 2)):
            st[k].pos[j][0], st[k].pos[j][1] = st[k].pos[i][0], st[k].pos[i][1]
            st[k].carry[i], st[k].carry[j] = 1 + (k % 5), 3
    env.set_state(st)
    obs, _ = env.observe()
    check(obs.clone(), "same cell")
    env.close()
