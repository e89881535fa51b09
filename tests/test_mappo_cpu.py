"""CPU suite: the MAPPO model / PPO loss / optimizer step against G7 (captured from the reference's MAPPOAgent with
closed-form weights), tolerance 1e-4 relative as BASELINE.json states; flat-bucket Adam vs torch.optim.Adam; and the
world_size-2 gloo data-parallel step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import _golden as G

RTOL = 1e-4   # BASELINE.json north_star: "PPO losses match within 1e-4 rtol"


def closed_form_weights(model):
    """Same formula as tests/golden/make_golden.py::closed_form_weights."""
    with torch.no_grad():
        for j, (name, p) in enumerate(model.named_parameters()):
            k = torch.arange(p.numel(), dtype=torch.float64)
            base = torch.sin(0.37 * k + 1.3 * j)
            if p.dim() >= 2:
                v = base * (0.5 / np.sqrt(p[0].numel()))
            elif name.endswith("weight"):
                v = 1.0 + 0.1 * base
            else:
                v = 0.05 * base
            p.copy_(v.reshape(p.shape).to(torch.float32))


def _close(a, b, rtol=RTOL, atol=1e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"max rel err {np.max(np.abs(a - b) / (np.abs(b) + 1e-12))}"


def _golden_batch():
    d, meta = G.load("ppo.npz")
    t = lambda k, dt=torch.float32: torch.tensor(d[k]).to(dt)
    return d, meta, t("obs"), t("merged"), t("act", torch.long), t("old_logp"), t("adv"), t("ret")


def test_model_forward_and_ppo_step_match_reference():
    from pmx import mappo
    torch.set_num_threads(2)
    d, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    model = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    assert [n for n, _ in model.named_parameters()] == meta["param_names"]
    assert sum(p.numel() for p in model.parameters()) == int(d["n_params"]) == 2634070      # SURVEY a20, smallCapture
    closed_form_weights(model)
    with torch.no_grad():
        _close(model.logits(obs).numpy(), d["logits"])
        _close(model.value(merged).numpy(), d["values"])
    learner = mappo.PPOLearner(model, lr=meta["lr"])
    loss, stats = mappo.ppo_loss(model, obs, merged, act, old_logp, adv, ret, meta["clip_eps"], meta["ent_coef"])
    _close(stats["pg"], d["pg"]); _close(stats["vl"], d["vl"]); _close(loss.item(), d["loss"])
    vals, logp, ent = model.evaluate(obs, merged, act)
    _close(logp.detach().numpy(), d["logp"]); _close(ent.detach().numpy(), d["entropy"])
    st = learner.update_minibatch(obs, merged, act, old_logp, adv, ret, meta["clip_eps"], meta["ent_coef"])
    _close(st["grad_norm"], d["grad_norm"])
    _close(float(learner.bucket.data.double().sum()), d["post_adam_sum"], rtol=1e-6)
    _close(float(learner.bucket.data.double().abs().sum()), d["post_adam_abs"], rtol=1e-6)


def sharpen_init(model):
    """Same transform as tests/golden/make_golden.py::sharpen_init (logits and values of O(1) from the seeded initialisation)."""
    with torch.no_grad():
        model.actor_head[3].weight.mul_(120.0)
        model.critic_head[2].weight.mul_(6.0)
        model.critic_projector[0].weight.mul_(2.5)
        for j, (name, p) in enumerate(model.named_parameters()):
            if p.dim() == 1 and name.endswith("bias"):
                p.add_(0.1 * torch.sin(0.37 * torch.arange(p.numel(), dtype=torch.float64) + 1.3 * j).to(p.dtype))


def reference_init_model(tag, shape, seed):
    """MAPPOAgent as the REFERENCE initialises it under torch.manual_seed(seed) (fixture G7b stores the seed and per-tensor
    checksums instead of the weights): the module list, the creation order and the init calls mirror
    pacman_mappo_resnet.py:97-158, so the same seed draws the same weights."""
    from pmx import mappo
    n_thr = torch.get_num_threads()
    torch.set_num_threads(1)               # as the generator: the QR behind nn.init.orthogonal_ rounds differently with more threads
    try:
        torch.manual_seed(seed)
        model = mappo.MAPPOAgent(shape, 5, 2)
    finally:
        torch.set_num_threads(n_thr)
    if tag == "sharp":
        sharpen_init(model)
    return model


INIT_FIXTURES = {"small": "ppo_init.npz", "blox": "ppo_init_blox.npz", "tiny": "ppo_init_tiny.npz"}   # G7b smallCapture, G7c bloxCapture (20 x 20), G7d tinyCapture


def _init_batch(tag, board="small"):
    d, meta = G.load(INIT_FIXTURES[board])
    t = lambda k, dt=torch.float32: torch.tensor(d[k]).to(dt)
    return (d, meta, t("obs"), t("merged"), t(f"{tag}_act", torch.long), t(f"{tag}_old_logp"), t(f"{tag}_adv"), t(f"{tag}_ret"))


@pytest.mark.parametrize("board", ["small", "blox", "tiny"])
@pytest.mark.parametrize("tag", ["init", "sharp"])
def test_reference_initialisation_fixture_fp32(tag, board):
    """G7b / G7c (the same on the 20 x 20 board the reference trains on): the reference's own seeded initialisation is reproduced weight for weight (per-tensor sums), and the float32 model
    gives the reference's logits, values, log-probabilities, entropies, losses and per-tensor gradient norms within 1e-4 on
    the paired batch (merged input k serves rows 2k and 2k + 1)."""
    from pmx import mappo
    torch.set_num_threads(2)
    d, meta, obs, merged, act, old_logp, adv, ret = _init_batch(tag, board)
    model = reference_init_model(tag, tuple(obs.shape[1:]), meta["seed"])
    assert [n for n, _ in model.named_parameters()] == meta["param_names"]
    # (the same draws; a LAPACK build or thread count that rounds the QR differently moves a sum in its 6th digit at most)
    _close([float(p.detach().double().sum()) for p in model.parameters()], d[f"{tag}_param_sum"], rtol=2e-5, atol=2e-5)
    _close([float(p.detach().double().abs().sum()) for p in model.parameters()], d[f"{tag}_param_abs"], rtol=2e-5, atol=2e-5)
    with torch.no_grad():
        _close(model.logits(obs).numpy(), d[f"{tag}_logits"])
        _close(model.value(merged).repeat_interleave(2).numpy(), d[f"{tag}_values"])
    loss, stats = mappo.ppo_loss(model, obs, merged, act, old_logp, adv, ret, meta["clip_eps"], meta["ent_coef"])
    _close(stats["pg"], d[f"{tag}_pg"]); _close(stats["vl"], d[f"{tag}_vl"]); _close(loss.item(), d[f"{tag}_loss"])
    _close(stats["entropy"], np.mean(d[f"{tag}_entropy"]))
    grads = torch.autograd.grad(loss, list(model.parameters()))
    _close([float(g.double().norm()) for g in grads], d[f"{tag}_grad_norms"], rtol=2e-4, atol=1e-7)
    _close(float(torch.linalg.vector_norm(torch.stack([g.norm() for g in grads]))), d[f"{tag}_grad_norm"])


def test_flat_bucket_adam_equals_torch_adam():
    from pmx import mappo
    torch.manual_seed(0)
    _, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    a = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    b = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    b.load_state_dict(a.state_dict())
    learner = mappo.PPOLearner(a, lr=2e-4)
    opt = torch.optim.Adam(b.parameters(), lr=2e-4, eps=1e-5)
    ema = [p.detach().clone() for p in b.parameters()]
    for step in range(3):
        learner.update_minibatch(obs, merged, act, old_logp, adv, ret)
        loss, _ = mappo.ppo_loss(b, obs, merged, act, old_logp, adv, ret, mappo.CLIP_EPS, mappo.ENT_COEF_START)
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), mappo.MAX_GRAD_NORM)
        opt.step()
        with torch.no_grad():
            for e, p in zip(ema, b.parameters()):
                e.mul_(mappo.EMA_DECAY).add_(p.data, alpha=1 - mappo.EMA_DECAY)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    sd = learner.ema_state_dict()
    for (n, _), e in zip(b.named_parameters(), ema):
        assert torch.allclose(sd[n], e, rtol=1e-5, atol=1e-7), n
    assert set(sd) == set(b.state_dict())


def test_canonicalize_action_and_schedule():
    from pmx import mappo
    d, _ = G.load("shaping.npz")
    assert [mappo.canonicalize_action(a, True) for a in range(5)] == list(d["action_map_red"])
    assert [mappo.canonicalize_action(a, False) for a in range(5)] == [0, 1, 2, 3, 4]
    t = torch.tensor([0, 1, 2, 3, 4])
    assert mappo.canonicalize_action(t, True).tolist() == list(d["action_map_red"])
    lr, ent, clip = mappo.schedule(0, 2000)
    assert (lr, ent, clip) == (2e-4, 0.02, 0.15)
    lr, ent, clip = mappo.schedule(2000, 2000)
    assert abs(lr - 4e-5) < 1e-12 and abs(ent - 0.004) < 1e-12 and clip == 0.1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from pmx import mappo
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    _, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    half = obs.shape[0] // 2
    sl = slice(rank * half, (rank + 1) * half)
    torch.manual_seed(1)
    model = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)       # same seed -> replicated weights
    learner = mappo.PPOLearner(model, lr=2e-4, world_size=world)
    for _ in range(2):
        learner.update_minibatch(obs[sl], merged[sl], act[sl], old_logp[sl], adv[sl], ret[sl])
    q.put((rank, learner.bucket.data.clone().numpy(), learner.ema.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """Two ranks, disjoint minibatch halves, one flat all-reduce per step: ranks stay bit-identical and equal a
    single process that averages the two per-shard gradients by hand (advantages normalised per local minibatch,
    as the reference does per minibatch, pacman_mappo_resnet.py:577)."""
    from pmx import mappo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict()
    for _ in range(2):
        r, data, ema = q.get(timeout=300)
        res[r] = (data, ema)
    for p in procs: p.join(60)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # single-process emulation
    torch.set_num_threads(2)
    _, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    half = obs.shape[0] // 2
    torch.manual_seed(1)
    model = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    learner = mappo.PPOLearner(model, lr=2e-4)
    for _ in range(2):
        grads = []
        for r in range(2):
            sl = slice(r * half, (r + 1) * half)
            learner.bucket.grad.zero_()
            loss, _ = mappo.ppo_loss(model, obs[sl], merged[sl], act[sl], old_logp[sl], adv[sl], ret[sl], mappo.CLIP_EPS, mappo.ENT_COEF_START)
            loss.backward()
            grads.append(learner.bucket.grad.clone())
        learner.bucket.grad.copy_((grads[0] + grads[1]) / 2)
        gn = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([p.grad for p in learner.bucket.params])))
        learner.bucket.grad.mul_(torch.clamp(mappo.MAX_GRAD_NORM / (gn + 1e-6), max=1.0))
        learner._adam_step()
    # thread counts (1 vs 2) change conv-backward summation order; Adam turns that into <= ~1e-2 of a 2e-4 step
    assert np.allclose(learner.bucket.data.numpy(), res[0][0], rtol=1e-4, atol=5e-6)


def test_paired_minibatch_loss_equals_expanded_loss():
    """ppo_loss with ONE merged critic input per env-tick pair == the reference's form with that input repeated per agent."""
    from pmx import mappo
    torch.manual_seed(5)
    shape, P = (8, 7, 20), 6
    model = mappo.MAPPOAgent(shape, 5, 2)
    obs = (torch.rand((2 * P,) + shape) < 0.2).float()
    merged = (torch.rand((P,) + shape) < 0.2).float()
    act = torch.randint(0, 5, (2 * P,))
    logp, adv, ret = -1.6 + 0.1 * torch.randn(2 * P), torch.randn(2 * P), torch.randn(2 * P)
    la, _ = mappo.ppo_loss(model, obs, merged, act, logp, adv, ret, 0.15, 0.02)
    ga = torch.autograd.grad(la, [p for p in model.parameters() if p.requires_grad], allow_unused=True)
    lb, _ = mappo.ppo_loss(model, obs, merged.repeat_interleave(2, dim=0), act, logp, adv, ret, 0.15, 0.02)
    gb = torch.autograd.grad(lb, [p for p in model.parameters() if p.requires_grad], allow_unused=True)
    assert torch.allclose(la, lb, rtol=1e-6, atol=1e-7)
    for x, y in zip(ga, gb):
        if x is not None:
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6)


class _StubEnv:
    """Stand-in for PmxVecEnv in host-side tests of VecMAPPOTrainer (update / opponent schedule): layout + zero observations."""

    def __init__(self, layout, n_envs):
        from pmx.layout import get_layout
        self.layout = get_layout(layout)
        self.obs_torch_dtype = torch.float32
        self.n_envs = n_envs

    def reset(self):
        return torch.zeros((self.n_envs, 4, 8, self.layout.height, self.layout.width)), None

    def close(self):
        pass


def _trainer_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from pmx import trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    N, T = 6, 4
    tr = trainer.VecMAPPOTrainer("tinyCapture", N, horizon=T, minibatch=16, device="cpu", seed=3, rank=rank, world_size=world,
                                 process_group=dist.group.WORLD, use_autocast=False, opponent="curriculum", curriculum_scale=0.005,
                                 total_updates=50, env=_StubEnv("tinyCapture", N))
    g = torch.Generator().manual_seed(100 + rank)                 # every rank has its OWN rollout data
    modes = []
    for u in range(3):
        tr.update_idx = (0, 3, 7)[u]                               # one update in each curriculum phase (thresholds 1 and 4)
        modes.append(tr._pick_opponent())
        tr.obs_buf.copy_((torch.rand(tr.obs_buf.shape, generator=g) < 0.2).float())
        tr.merged_buf.copy_((torch.rand(tr.merged_buf.shape, generator=g) < 0.2).float())
        tr.act_buf.copy_(torch.randint(0, 5, tr.act_buf.shape, generator=g))
        tr.logp_buf.copy_(-1.6 + 0.1 * torch.randn(tr.logp_buf.shape, generator=g))
        tr.adv_buf.copy_(torch.randn(tr.adv_buf.shape, generator=g))
        tr.ret_buf.copy_(torch.randn(tr.ret_buf.shape, generator=g))
        tr.update()
    q.put((rank, modes, tr.learner.bucket.data.clone().numpy(), tr.learner.ema.clone().numpy(), int(tr.stats["optimizer_steps"])))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_update_two_ranks_gloo_same_modes_same_weights():
    """VecMAPPOTrainer.update under torch.distributed (gloo, 2 ranks, CPU tensors, different data per rank): the opponent
    mode / side sequence of the curriculum is the same on every rank (it must be: a rank that plays self-play does an extra
    forward per tick while the others wait at the all-reduce) and the weights and the EMA stay bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = {}
    for _ in range(2):
        r, modes, data, ema, steps = q.get(timeout=600)
        res[r] = (modes, data, ema, steps)
    for p in procs: p.join(60)
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert res[0][0][0][0] == "random" and res[0][0][1][0] in ("random", "baseline") and res[0][0][2][0] in ("self", "pool", "random", "baseline")
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][3] == res[1][3] == 3 * (4 * 6 * 2 // 16)


def test_curriculum_phases_and_pool_draws_on_the_host():
    """_pick_opponent (pacman_mappo_resnet.py:396-438): randomTeam only in the first phase, randomTeam / baselineTeam in the
    second, self-play / pool / bots with both colours in the third; the pool grows every OPPONENT_UPDATE_FREQ updates."""
    from pmx import trainer
    tr = trainer.VecMAPPOTrainer("tinyCapture", 4, horizon=2, minibatch=8, device="cpu", seed=1, use_autocast=False,
                                 opponent="curriculum", curriculum_scale=0.1, env=_StubEnv("tinyCapture", 4))
    seen = {1: set(), 2: set(), 3: set()}
    sides = set()
    for phase, idx in ((1, 10), (2, 50), (3, 200)):
        for _ in range(200):
            tr.update_idx = idx
            mode, red = tr._pick_opponent()
            seen[phase].add(mode)
            if mode in ("self", "pool"):
                sides.add(red)
            else:
                assert red is False
    assert seen[1] == {"random"} and seen[2] == {"random", "baseline"} and seen[3] == {"self", "pool", "random", "baseline"}
    assert sides == {True, False}
    n0 = len(tr.opponent_pool)
    for k, v in (("obs_buf", 0.0), ("merged_buf", 0.0)):
        getattr(tr, k).fill_(v)
    tr.adv_buf.normal_(); tr.ret_buf.normal_(); tr.logp_buf.fill_(-1.6)
    tr.update_idx = 0
    tr.update()                                                    # update 0 is a multiple of the snapshot frequency
    assert len(tr.opponent_pool) == n0 + 1 and tr.update_idx == 1
    tr.opponent_mode = "pool"
    mode, _ = tr._pick_opponent()
    assert mode == "pool"
    sd = tr.opponent_model.state_dict()
    assert any(all(torch.equal(sd[k], snap[k]) for k in snap) for snap in tr.opponent_pool)


def test_full_checkpoint_round_trip_continues_identically(tmp_path):
    """save_full -> load_full into a fresh trainer: the next update (same rollout buffers) gives bit-identical weights,
    moments, EMA, pool and random draws; the file loads with weights_only=True (tensors and numbers only)."""
    from pmx import trainer

    def make():
        return trainer.VecMAPPOTrainer("tinyCapture", 4, horizon=3, minibatch=8, device="cpu", seed=2, use_autocast=False,
                                       opponent="pool", total_updates=40, env=_StubEnv("tinyCapture", 4))

    def fill(tr, seed):
        g = torch.Generator().manual_seed(seed)
        tr.obs_buf.copy_((torch.rand(tr.obs_buf.shape, generator=g) < 0.2).float())
        tr.merged_buf.copy_((torch.rand(tr.merged_buf.shape, generator=g) < 0.2).float())
        tr.act_buf.copy_(torch.randint(0, 5, tr.act_buf.shape, generator=g))
        tr.logp_buf.copy_(-1.6 + 0.1 * torch.randn(tr.logp_buf.shape, generator=g))
        tr.adv_buf.copy_(torch.randn(tr.adv_buf.shape, generator=g)); tr.ret_buf.copy_(torch.randn(tr.ret_buf.shape, generator=g))
    a = make()
    fill(a, 1); a.update(); a._pick_opponent()
    path = str(tmp_path / "full.pt")
    a.save_full(path)
    torch.load(path, weights_only=True)                            # nothing in the file needs the unpickler
    b = make()
    b.load_full(path)
    assert b.update_idx == a.update_idx == 1 and b.learner.step_count == a.learner.step_count
    for tr in (a, b):
        fill(tr, 2); tr.update()
    for x, y in ((a.learner.bucket.data, b.learner.bucket.data), (a.learner.ema, b.learner.ema),
                 (a.learner.exp_avg, b.learner.exp_avg), (a.learner.exp_avg_sq, b.learner.exp_avg_sq)):
        assert torch.equal(x, y)
    assert [a._pick_opponent() for _ in range(5)] == [b._pick_opponent() for _ in range(5)]
    assert len(a.opponent_pool) == len(b.opponent_pool)
    c = trainer.VecMAPPOTrainer("tinyCapture", 4, horizon=3, minibatch=8, device="cpu", seed=2, use_autocast=False,
                                total_updates=41, env=_StubEnv("tinyCapture", 4))
    with pytest.raises(ValueError):
        c.load_full(path)                                          # another schedule length: refuse


def test_ema_checkpoint_loads_into_a_fresh_agent(tmp_path):
    """save_ema writes what the reference saves (pacman_mappo_resnet.py:647-651): a state_dict of the EMA weights with the
    reference's parameter names, loadable into a fresh MAPPOAgent."""
    from pmx import mappo, trainer
    tr = trainer.VecMAPPOTrainer("tinyCapture", 4, horizon=2, minibatch=8, device="cpu", seed=4, use_autocast=False,
                                 env=_StubEnv("tinyCapture", 4))
    tr.adv_buf.normal_(); tr.ret_buf.normal_(); tr.logp_buf.fill_(-1.6)
    tr.update()
    path = str(tmp_path / "ema.pt")
    tr.save_ema(path)
    sd = torch.load(path, weights_only=True)
    fresh = mappo.MAPPOAgent(tr.obs_shape, 5, 2)
    fresh.load_state_dict(sd, strict=True)
    flat = torch.cat([p.detach().reshape(-1) for p in fresh.parameters()])
    assert torch.equal(flat, tr.learner.ema)
    assert not torch.equal(flat, tr.learner.bucket.data)           # one step at EMA 0.995: the EMA lags the weights
