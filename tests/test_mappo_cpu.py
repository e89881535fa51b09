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
