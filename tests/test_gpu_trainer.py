"""GPU tests of the training side: shaping from the compact self plane vs the oracle's observation-based restatement,
rollout-buffer consistency, GAE on the rollout vs the oracle, PPO loss parity on the GPU (fp32, 1e-4 rtol), and a
short end-to-end training run."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_shaping_from_agent_words_matches_observation_shaping():
    import pmx
    from pmx import trainer
    lay = pmx.get_layout("smallCapture")
    N = 512
    env = pmx.PmxVecEnv(lay, N, length=300, auto_reset=False, seed=3)
    _, meta = G.load("scen_small_random.npz")
    d, _ = G.load("scen_small_random.npz")
    K = len(d["actions"])
    states = [pmx.make_state(d["in_pos"][k % K], d["in_dir"][k % K], d["in_pac"][k % K], d["in_scared"][k % K],
                             d["in_carry"][k % K], d["in_ret"][k % K], d["in_food"][k % K], d["in_caps"][k % K],
                             d["in_score"][k % K], 0, lay.height) for k in range(N)]
    env.set_state(states)
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
    obs0, _, _, info = env.step(a)
    obs0 = obs0.clone(); w0 = info["agent"].clone()
    for t in range(12):
        a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
        obs1, _, _, info = env.step(a)
        w1 = info["agent"].clone()
        shp = trainer.shaping_from_agent_words(w0, w1).cpu().numpy()
        o0, o1 = obs0.cpu().numpy(), obs1.cpu().numpy()
        for e in range(0, N, 5):
            for i in range(4):
                c0 = O.canonicalize_obs(o0[e, i]) if i in (0, 2) else o0[e, i]
                c1 = O.canonicalize_obs(o1[e, i]) if i in (0, 2) else o1[e, i]
                assert O.shaping(c0, c1) == shp[e, i], (t, e, i)
        obs0, w0 = obs1.clone(), w1
    env.close()


def test_canonicalize_and_merge_torch_ops_match_golden():
    from pmx import trainer
    d, meta = G.load("shaping.npz")
    tr, _ = G.load(meta["traj"])
    for j, t in enumerate(d["ticks"]):
        o = torch.tensor(tr["obs"][t]).cuda().float()
        canon = trainer.canonicalize_obs(o)
        assert (canon.cpu().numpy() == d["canon_red"][j]).all()
        assert (trainer.merge_obs(o[1:2], o[3:4])[0].cpu().numpy() == d["merged_blue"][j]).all()
        assert (trainer.merge_obs(canon[0:1], canon[2:3])[0].cpu().numpy() == d["merged_red"][j]).all()


def test_ppo_loss_on_gpu_fp32_matches_reference():
    from pmx import mappo
    from test_mappo_cpu import closed_form_weights, _golden_batch, _close
    d, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    model = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2)
    closed_form_weights(model)
    model = model.cuda()
    learner = mappo.PPOLearner(model, lr=meta["lr"])
    c = lambda x: x.cuda()
    loss, stats = mappo.ppo_loss(model, c(obs), c(merged), c(act), c(old_logp), c(adv), c(ret), meta["clip_eps"], meta["ent_coef"])
    _close(stats["pg"].cpu(), d["pg"]); _close(stats["vl"].cpu(), d["vl"]); _close(loss.item(), d["loss"])
    st = learner.update_minibatch(c(obs), c(merged), c(act), c(old_logp), c(adv), c(ret), meta["clip_eps"], meta["ent_coef"])
    _close(st["grad_norm"].cpu(), d["grad_norm"])
    _close(float(learner.bucket.data.double().sum()), d["post_adam_sum"], rtol=1e-5)


@pytest.mark.parametrize("board", ["small", "blox", "tiny"])
def test_ppo_loss_on_gpu_fp32_matches_the_seeded_initialisation_fixtures(board):
    """BASELINE's 1e-4 on the GPU in float32 for the reference's own initialisation (fixtures G7b / G7c / G7d, "sharp" variant: logits and
    values of O(1)) on all three boards: library float32 ops for the network, the one-launch loss kernel (logits, values, the three losses and the entropy within
    1e-4 / 2e-4), and the gathered gradient's global norm within 1e-3.  Paired minibatch: merged input k serves rows 2k and 2k + 1."""
    from pmx import mappo
    from test_mappo_cpu import _init_batch, reference_init_model, _close
    d, meta, obs, merged, act, old_logp, adv, ret = _init_batch("sharp", board)
    model = reference_init_model("sharp", tuple(obs.shape[1:]), meta["seed"]).cuda()
    learner = mappo.PPOLearner(model, lr=meta["lr"])
    c = lambda x: x.cuda()
    with torch.no_grad():
        _close(model.logits(c(obs)).cpu().numpy(), d["sharp_logits"], rtol=2e-4, atol=2e-5)
        _close(model.value(c(merged)).repeat_interleave(2).cpu().numpy(), d["sharp_values"], rtol=2e-4, atol=2e-5)
    loss, stats = mappo.ppo_loss(model, c(obs), c(merged), c(act), c(old_logp), c(adv), c(ret), meta["clip_eps"], meta["ent_coef"])
    _close(float(stats["pg"]), d["sharp_pg"], rtol=2e-4, atol=2e-6); _close(float(stats["vl"]), d["sharp_vl"]); _close(loss.item(), d["sharp_loss"])
    _close(float(stats["entropy"]), np.mean(d["sharp_entropy"]))
    learner._backward_into_bucket(loss)
    # the gradient passes through MIOpen's float32 backward convolutions (measured: 6e-5 / 1.1e-4 / 4.2e-4 relative on small / tiny /
    # blox); the same model's 50 per-tensor gradient norms are held to 2e-4 by the CPU test of these fixtures
    _close(float(learner.bucket.grad.double().norm()), d["sharp_grad_norm"], rtol=1e-3)


@pytest.mark.parametrize("opponent,dtype,algorithm", [("random", "bfloat16", "mappo"), ("self", "uint8", "mappo"), ("random", "bfloat16", "ippo"), ("baseline", "bfloat16", "mappo")])
def test_rollout_gae_update_end_to_end(opponent, dtype, algorithm):
    import pmx
    from pmx import trainer
    tr = trainer.VecMAPPOTrainer("smallCapture", n_envs=256, horizon=12, minibatch=512, epochs=2, obs_dtype=dtype,
                                 seed=5, length=20, opponent=opponent, algorithm=algorithm)
    for u in range(2):
        tr.rollout()
        tr.compute_gae()
        # GAE of the rollout buffers against the oracle, bit-exact, for a sample of series
        rew, val, done = tr.rew_buf.cpu().numpy(), tr.val_buf.cpu().numpy(), tr.done_buf.cpu().numpy()
        adv, ret, last = tr.adv_buf.cpu().numpy(), tr.ret_buf.cpu().numpy(), tr.last_value.cpu().numpy()   # last [N,2]
        for e in range(0, 256, 37):
            for i in range(2):
                a, r = O.gae(rew[:, e, i], val[:, e, i], done[:, e, i], float(last[e, i]), 0.99, 0.95)
                assert a.tobytes() == adv[:, e, i].tobytes() and r.tobytes() == ret[:, e, i].tobytes()
        if u == 1:
            assert done.sum() > 0                                # length 20 -> every episode ends at tick 21
        # buffers: both learners share value and done; merged plane 4 is zero, plane 1 holds both learners
        assert torch.equal(tr.done_buf[..., 0], tr.done_buf[..., 1])
        assert torch.equal(tr.val_buf[..., 0], tr.val_buf[..., 1]) == (algorithm == "mappo")
        if algorithm == "mappo":                                 # IPPO's critic reads the agents' own planes: no merged input is written
            m = tr.merged_buf.float()
            assert float(m[:, :, 4].abs().sum()) == 0
            assert torch.equal((m[:, :, 1] > 0).sum((-1, -2)) >= 1, torch.ones_like(m[:, :, 1, 0, 0], dtype=torch.bool))
        tr.update()
        s = tr.stats
        for k in ("pg", "vl", "entropy", "loss", "grad_norm"):
            assert torch.isfinite(s[k]).all(), k
        assert s["optimizer_steps"] == 2 * (12 * 256 * 2 // 512)
    assert 0.5 < float(tr.stats["entropy"]) <= float(np.log(5)) + 1e-3
    tr.env.close()


def test_graph_captured_step_equals_eager_step():
    """hipGraph replay of the optimizer step against the eager step: same data, fp32, three steps."""
    from pmx import mappo
    from test_mappo_cpu import _golden_batch
    d, meta, obs, merged, act, old_logp, adv, ret = _golden_batch()
    c = lambda x: x.cuda()
    torch.manual_seed(3)
    a = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2).cuda()
    b = mappo.MAPPOAgent(tuple(obs.shape[1:]), 5, 2).cuda()
    b.load_state_dict(a.state_dict())
    la, lb = mappo.PPOLearner(a, lr=2e-4), mappo.PPOLearner(b, lr=2e-4)
    lb.capture(obs.shape[0], obs.shape[1:], torch.float32)
    assert torch.equal(la.bucket.data, lb.bucket.data)            # capture restores the optimizer state
    for k in range(3):
        sa = la.update_minibatch(c(obs), c(merged), c(act), c(old_logp), c(adv), c(ret), 0.15, 0.02)
        sb = lb.update_minibatch_graph(c(obs), c(merged), c(act), c(old_logp), c(adv), c(ret), 0.15, 0.02)
        assert torch.allclose(sa["loss"], sb["loss"], rtol=1e-4, atol=1e-6), k
        assert torch.allclose(sa["grad_norm"], sb["grad_norm"], rtol=1e-3), k
    assert torch.allclose(la.bucket.data, lb.bucket.data, rtol=1e-3, atol=2e-5)
    assert torch.allclose(la.ema, lb.ema, rtol=1e-3, atol=2e-5)


def test_graph_replay_bf16_stays_finite_past_230_replays():
    """Regression for the ROCm graph packet-capture bug: with the packet capture left on, the 233rd
    replay of the bf16 optimizer-step graph, interleaved with ordinary launches, returned an infinite gradient norm."""
    from pmx import mappo
    assert mappo.PPOLearner.graph_replay_safe()
    torch.manual_seed(0)
    shape, B = (8, 11, 14), 256
    model = mappo.MAPPOAgent(shape, 5, 2).cuda()
    L = mappo.PPOLearner(model, autocast_dtype=torch.bfloat16)
    L.capture(B, shape, torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(1)
    norms = []
    for it in range(300):
        obs = (torch.rand((B,) + shape, device="cuda", generator=g) < 0.2).to(torch.bfloat16)
        mg = (torch.rand((B,) + shape, device="cuda", generator=g) < 0.2).to(torch.bfloat16)
        act = torch.randint(0, 5, (B,), device="cuda", generator=g)
        logp = -1.6 + 0.05 * torch.randn(B, device="cuda", generator=g)
        st = L.update_minibatch_graph(obs, mg, act, logp, torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g))
        norms.append(st["grad_norm"].clone())
    norms = torch.stack(norms)
    assert bool(torch.isfinite(norms).all()), torch.nonzero(~torch.isfinite(norms)).flatten().tolist()
    assert bool(torch.isfinite(L.bucket.data).all())


def test_graph_replay_survives_larger_tower_calls_after_capture():
    """A captured bf16 step replays a fused-tower backward that writes through the scratch pointer recorded at capture.  That
    scratch must belong to the graph: larger inference and training calls made AFTER the capture (they used to replace a
    module-global grow-only buffer and hand its old block back to the allocator) must not disturb later replays.  The replayed
    learner is compared step by step with a twin that runs the same steps eagerly and makes the same interleaved calls."""
    from pmx import actor_tower, mappo
    assert mappo.PPOLearner.graph_replay_safe()
    shape, B = (8, 11, 14), 512
    torch.manual_seed(5)
    a = mappo.MAPPOAgent(shape, 5, 2).cuda()
    b = mappo.MAPPOAgent(shape, 5, 2).cuda()
    b.load_state_dict(a.state_dict())
    la, lb = mappo.PPOLearner(a, autocast_dtype=torch.bfloat16), mappo.PPOLearner(b, autocast_dtype=torch.bfloat16)
    lb.capture(B, shape, torch.uint8, merged_batch=B // 2)
    g = torch.Generator(device="cuda").manual_seed(6)
    big = (torch.rand((6000,) + shape, device="cuda", generator=g) < 0.2).to(torch.uint8)
    junk = []
    for k in range(6):
        obs = (torch.rand((B,) + shape, device="cuda", generator=g) < 0.2).to(torch.uint8)
        mg = (torch.rand((B // 2,) + shape, device="cuda", generator=g) < 0.2).to(torch.uint8)
        act = torch.randint(0, 5, (B,), device="cuda", generator=g)
        logp = -1.6 + 0.05 * torch.randn(B, device="cuda", generator=g)
        adv, ret = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
        sa = la.update_minibatch(obs, mg, act, logp, adv, ret)
        sb = lb.update_minibatch_graph(obs, mg, act, logp, adv, ret)
        assert torch.allclose(sa["loss"], sb["loss"], rtol=2e-3, atol=2e-4), (k, sa["loss"], sb["loss"])
        assert torch.allclose(sa["grad_norm"], sb["grad_norm"], rtol=2e-2), (k, sa["grad_norm"], sb["grad_norm"])
        # between replays: a larger inference call and a larger differentiated call through the same tower kernels, on both
        # models, plus allocations that would land in any block the capture's scratch had given back
        for m in (a, b):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                m.logits(big)
            feat = actor_tower.actor_tower(m.actor_backbone, big[:3000])
            torch.autograd.grad(feat.float().square().mean(), actor_tower._tower_params(m.actor_backbone))
        junk.append(torch.full((64 << 20,), float(k), device="cuda"))
    cos = torch.nn.functional.cosine_similarity(la.bucket.data - la.ema, lb.bucket.data - lb.ema, dim=0)
    assert float(cos) > 0.98, float(cos)
    assert bool(torch.isfinite(lb.bucket.data).all())


def test_flat_bf16_step_tracks_autocast_step():
    """PPOLearner.enable_bf16_flat (one flat bfloat16 weight copy, flat gradient) against the autocast step on the same data."""
    from pmx import mappo
    torch.manual_seed(2)
    shape, B = (8, 11, 14), 256
    a = mappo.MAPPOAgent(shape, 5, 2).cuda()
    b = mappo.MAPPOAgent(shape, 5, 2).cuda()
    b.load_state_dict(a.state_dict())
    la = mappo.PPOLearner(a, autocast_dtype=torch.bfloat16)
    lb = mappo.PPOLearner(b, autocast_dtype=torch.bfloat16)
    lb.enable_bf16_flat()
    g = torch.Generator(device="cuda").manual_seed(1)
    for k in range(4):
        obs = (torch.rand((B,) + shape, device="cuda", generator=g) < 0.2).to(torch.bfloat16)
        mg = (torch.rand((B // 2,) + shape, device="cuda", generator=g) < 0.2).to(torch.bfloat16)
        act = torch.randint(0, 5, (B,), device="cuda", generator=g)
        logp = -1.6 + 0.05 * torch.randn(B, device="cuda", generator=g)
        adv, ret = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
        sa = la.update_minibatch(obs, mg, act, logp, adv, ret)
        sb = lb.update_minibatch(obs, mg, act, logp, adv, ret)
        assert torch.allclose(sa["loss"], sb["loss"], rtol=3e-2, atol=3e-3), (k, sa["loss"], sb["loss"])
        assert torch.allclose(sa["grad_norm"], sb["grad_norm"], rtol=0.15), (k, sa["grad_norm"], sb["grad_norm"])
    # Adam moves every weight by about lr per step whatever the gradient scale, so compare the direction of the total update
    da, db = la.bucket.data - la.ema, lb.bucket.data - lb.ema
    cos = torch.nn.functional.cosine_similarity(da, db, dim=0)
    assert float(cos) > 0.8, float(cos)
    assert torch.equal(lb._w16.detach().float(), lb.bucket.data.to(torch.bfloat16).float())


@pytest.mark.parametrize("rows,C", [(154 * 257, 96), (154 * 64, 32), (70000, 128), (5000, 8), (4099, 256)])
def test_column_sum_kernel_matches_torch(rows, C):
    """pmx_colsum_bf16 (bias gradients of the token linears) against a float64 sum of the same bfloat16 values."""
    from pmx import mappo
    torch.manual_seed(rows + C)
    x = (torch.randn(rows, C, device="cuda") * 0.7 + 0.1).to(torch.bfloat16)
    got = mappo.column_sums(x.view(rows // 1, 1, C) if rows % 2 else x)
    ref = x.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (C,)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(x.double().abs().sum(0).max())


def test_unpaired_minibatches_still_run():
    """paired_minibatches=False is the reference's independent shuffle of agent samples."""
    from pmx import trainer
    tr = trainer.VecMAPPOTrainer("tinyCapture", 128, horizon=8, minibatch=256, obs_dtype="bfloat16", opponent="random", paired_minibatches=False)
    assert not tr.paired
    s = tr.train_update()
    assert torch.isfinite(s["loss"]).all() and s["optimizer_steps"] == 3 * (8 * 128 * 2 // 256)
    tr.env.close()


def test_graph_trainer_bf16_updates():
    """The trainer with the graph-replayed bf16 optimizer step: three full updates stay finite and learn the same kind of
    statistics as the eager loop."""
    from pmx import trainer
    tr = trainer.VecMAPPOTrainer("smallCapture", 256, horizon=12, minibatch=512, obs_dtype="bfloat16", opponent="random", use_graph=True)
    for u in range(3):
        s = tr.train_update()
        for k in ("pg", "vl", "entropy", "loss", "grad_norm"):
            assert torch.isfinite(s[k]).all(), (u, k)
    assert 0.5 < float(tr.stats["entropy"]) <= float(np.log(5)) + 1e-3
    tr.env.close()


def test_evaluate_vs_bots_runs():
    from pmx import mappo, trainer
    torch.manual_seed(0)
    model = mappo.MAPPOAgent((8, 7, 20), 5, 2).cuda()
    mean, std, wr = trainer.evaluate_vs_bots(model, num_episodes=2, layout_file="tinyCapture", teams=("randomTeam", "baselineTeam"), length=30)
    assert np.isfinite(mean) and np.isfinite(std) and 0.0 <= wr <= 1.0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_fused_add_layernorm32_matches_torch(dtype, tol):
    """pmx_ln32_forward/backward against torch.nn.functional.layer_norm(x + a) in float32."""
    from pmx import mappo
    torch.manual_seed(0)
    S, B = 154, 257
    ln = torch.nn.LayerNorm(32).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(32).cuda() * 0.3 + 1.0); ln.bias.copy_(torch.randn(32).cuda() * 0.2)
    x = torch.randn(S, B, 32, device="cuda").to(dtype).requires_grad_(True)
    a = torch.randn(S, B, 32, device="cuda").to(dtype).requires_grad_(True)
    g = torch.randn(S, B, 32, device="cuda").to(dtype)
    y = mappo.add_layer_norm_small(x, a, ln)
    y.backward(g)
    got = (y.float(), x.grad.float(), a.grad.float(), ln.weight.grad.clone(), ln.bias.grad.clone())
    ln.zero_grad()
    x2 = x.detach().float().requires_grad_(True); a2 = a.detach().float().requires_grad_(True)
    y2 = torch.nn.functional.layer_norm(x2 + a2, (32,), ln.weight, ln.bias, ln.eps)
    y2.backward(g.float())
    ref = (y2, x2.grad, a2.grad, ln.weight.grad, ln.bias.grad)
    for name, u, v in zip(("y", "dx", "da", "dw", "db"), got, ref):
        scale = float(v.abs().max()) + 1e-6
        assert float((u - v).abs().max()) <= tol * scale * (8 if name in ("dw", "db") and dtype == torch.bfloat16 else 1), name


@pytest.mark.parametrize("dtype,tol,hw,cl", [(torch.float32, 3e-5, (11, 14), False), (torch.bfloat16, 3e-2, (11, 14), False),
                                              (torch.float32, 3e-5, (20, 20), False), (torch.float32, 3e-5, (32, 32), False),
                                              (torch.bfloat16, 3e-2, (11, 14), True), (torch.bfloat16, 3e-2, (20, 20), True),
                                              (torch.bfloat16, 3e-2, (32, 32), True)])
def test_fused_groupnorm_gelu_matches_torch(dtype, tol, hw, cl):
    """pmx_gn8_gelu_forward/backward against gelu(group_norm(h) + res) computed by torch in float32, with and without residual."""
    from pmx import mappo
    torch.manual_seed(1)
    B, C = 37, 32
    gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C).cuda() * 0.3 + 1.0); gn.bias.copy_(torch.randn(C).cuda() * 0.2)
    for with_res in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        h = (torch.randn(B, C, *hw, device="cuda") * 1.5 + 0.3).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
        r = torch.randn(B, C, *hw, device="cuda").to(dtype).contiguous(memory_format=fmt).requires_grad_(True) if with_res else None
        g = torch.randn(B, C, *hw, device="cuda").to(dtype).contiguous(memory_format=fmt)
        gn.zero_grad()
        y = mappo.group_norm_gelu(h, r, gn)
        y.backward(g)
        got = [y.float(), h.grad.float(), gn.weight.grad.clone(), gn.bias.grad.clone()] + ([r.grad.float()] if with_res else [])
        gn.zero_grad()
        h2 = h.detach().float().requires_grad_(True)
        r2 = r.detach().float().requires_grad_(True) if with_res else None
        z = torch.nn.functional.group_norm(h2, 4, gn.weight, gn.bias, gn.eps)
        y2 = torch.nn.functional.gelu(z + r2 if with_res else z)
        y2.backward(g.float())
        ref = [y2, h2.grad, gn.weight.grad, gn.bias.grad] + ([r2.grad] if with_res else [])
        for name, u, v in zip(("y", "dh", "dw", "db", "dres"), got, ref):
            scale = float(v.abs().max()) + 1e-6
            mult = 8 if name in ("dw", "db") and dtype == torch.bfloat16 else 1
            assert float((u - v).abs().max()) <= tol * scale * mult, (name, with_res, float((u - v).abs().max()), scale)


@pytest.mark.parametrize("S,B", [(154, 37), (140, 5), (400, 9), (33, 3), (1024, 2), (160, 3), (1, 2), (17, 1)])
def test_mfma_attention_forward_matches_torch(S, B):
    """pmx_attn8_forward against softmax(q k^T / sqrt(8)) v computed by torch in float32 from the same bf16 inputs.
    Random (asymmetric) data: a transposed or permuted fragment layout cannot pass."""
    from pmx import mappo
    torch.manual_seed(S)
    qkv = (torch.randn(S, B, 96, device="cuda") * 1.7).to(torch.bfloat16)
    out, lse = mappo.attention8_forward(qkv, want_lse=True)
    q, k, v = qkv.float().chunk(3, dim=-1)
    q, k, v = (t.reshape(S, B, 4, 8).permute(1, 2, 0, 3) for t in (q, k, v))          # [B, h, S, d]
    sc = torch.matmul(q, k.transpose(-1, -2)) / 8 ** 0.5
    ref = torch.matmul(torch.softmax(sc, -1), v).permute(2, 0, 1, 3).reshape(S, B, 32)
    err = float((out.float() - ref).abs().max())
    assert err <= 2e-2 * (float(ref.abs().max()) + 1e-6), err                          # bf16 probabilities / outputs
    assert float((lse - torch.logsumexp(sc, -1)).abs().max()) <= 2e-3


@pytest.mark.parametrize("amp,shift,far", [(4.0, -25.0, 50.0), (1.5, -5.0, 3.0)])
def test_mfma_attention_forward_with_large_and_shifted_scores(amp, shift, far):
    """The forward kernel exponentiates against a reference fixed per query before the loop over the keys: the bound |q| max |k|
    while that is small enough for every row's largest probability to stay a normal number (second case: rows whose scores ALL
    sit far below the bound -- a common component in every key, as an in-projection bias produces), the exact largest score found
    in a first pass otherwise (first case: scores of +-50 .. 300).  Neither may overflow or underflow."""
    from pmx import mappo
    torch.manual_seed(77)
    S, B = 154, 6
    x = torch.randn(S, B, 96, device="cuda") * amp
    x[:, :, 32:64] += shift * torch.sign(x[:1, :, 0:32])           # every key carries a large component opposed to the FIRST query's signs
    qkv = x.to(torch.bfloat16)
    out, lse = mappo.attention8_forward(qkv, want_lse=True)
    q, k, v = qkv.float().chunk(3, dim=-1)
    q, k, v = (t.reshape(S, B, 4, 8).permute(1, 2, 0, 3) for t in (q, k, v))
    sc = torch.matmul(q, k.transpose(-1, -2)) / 8 ** 0.5
    assert float(sc[:, :, 0].max()) < -far and float(sc.max()) > far           # rows far below zero and rows far above exist
    ref = torch.matmul(torch.softmax(sc, -1), v).permute(2, 0, 1, 3).reshape(S, B, 32)
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all())
    assert float((out.float() - ref).abs().max()) <= 2e-2 * (float(ref.abs().max()) + 1e-6)
    assert float(((lse - torch.logsumexp(sc, -1)).abs() / (1.0 + torch.logsumexp(sc, -1).abs())).max()) <= 2e-3


@pytest.mark.parametrize("S", [154, 400])
def test_mfma_attention_forward_rows_one_key_dominates(S):
    """Rows whose probability mass sits on ONE key (score gap ~20 nats), at magnitudes where the kernel exponentiates against the
    head's norm bound rather than the row maximum: no probability is exactly 1 there, so the log-sum-exp must come from the
    unrounded probabilities (a bf16-rounded dominant term would be off by up to 2^-9 = 2e-3 relative, i.e. in the log) and the output
    must still be the dominant key's value row."""
    from pmx import mappo
    torch.manual_seed(5)
    B = 4
    x = torch.randn(S, B, 96, device="cuda") * 0.3
    tgt = torch.randint(0, 8, (S,), device="cuda")                  # query i looks at key tgt[i], one of the first eight
    code = torch.eye(8, device="cuda") * 8.5                        # those keys carry orthogonal codes, the others only noise
    for h in range(4):
        x[:8, :, 32 + 8 * h:40 + 8 * h] += code[:, None, :]
        x[:, :, 8 * h:8 * h + 8] += code[tgt][:, None, :]             # queries carry their target's code
    qkv = x.to(torch.bfloat16)
    out, lse = mappo.attention8_forward(qkv, want_lse=True)
    q, k, v = qkv.float().chunk(3, dim=-1)
    q, k, v = (t.reshape(S, B, 4, 8).permute(1, 2, 0, 3) for t in (q, k, v))
    sc = torch.matmul(q.double(), k.double().transpose(-1, -2)) / 8 ** 0.5
    top2 = sc.topk(2, dim=-1).values
    assert float((top2[..., 0] - top2[..., 1]).median()) > 8.0           # most rows are dominated by one key
    ref_lse = torch.logsumexp(sc, -1)
    assert float((lse.double() - ref_lse).abs().max()) <= 5e-4, float((lse.double() - ref_lse).abs().max())
    ref = torch.matmul(torch.softmax(sc, -1), v.double()).permute(2, 0, 1, 3).reshape(S, B, 32)
    assert float((out.double() - ref).abs().max()) <= 2e-2 * (float(ref.abs().max()) + 1e-6)


@pytest.mark.parametrize("S,B", [(154, 19), (140, 5), (400, 4), (33, 3), (640, 2), (397, 3)])
def test_mfma_attention_backward_matches_torch(S, B):
    """pmx_attn8_backward against torch autograd of softmax(q k^T / sqrt(8)) v in float32 on the same bf16 inputs."""
    from pmx import mappo
    torch.manual_seed(S + 1)
    qkv = (torch.randn(S, B, 96, device="cuda") * 1.3).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(S, B, 32, device="cuda").to(torch.bfloat16)
    out = mappo.attention8(qkv)
    out.backward(g)
    got = qkv.grad.float()
    x = qkv.detach().float().requires_grad_(True)
    q, k, v = x.chunk(3, dim=-1)
    q, k, v = (t.reshape(S, B, 4, 8).permute(1, 2, 0, 3) for t in (q, k, v))
    ref_out = torch.matmul(torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / 8 ** 0.5, -1), v).permute(2, 0, 1, 3).reshape(S, B, 32)
    ref_out.backward(g.float())
    ref = x.grad
    for name, sl in (("dq", slice(0, 32)), ("dk", slice(32, 64)), ("dv", slice(64, 96))):
        err = float((got[..., sl] - ref[..., sl]).abs().max())
        assert err <= 3e-2 * (float(ref[..., sl].abs().max()) + 1e-6), (name, err, float(ref[..., sl].abs().max()))


def test_evaluate_vectorized_runs_and_random_policy_loses_to_baseline():
    from pmx import mappo, trainer
    torch.manual_seed(0)
    model = mappo.MAPPOAgent((8, 11, 14), 5, 2).cuda()
    mean, std, wr = trainer.evaluate_vectorized(model, layout="smallCapture", n_envs=256, opponent="baseline", length=120)
    assert np.isfinite(mean) and np.isfinite(std) and wr < 0.2          # an untrained argmax policy does not beat the reflex bots
    mean_r, _, wr_r = trainer.evaluate_vectorized(model, layout="smallCapture", n_envs=256, opponent="random", length=60)
    assert np.isfinite(mean_r) and 0.0 <= wr_r <= 1.0


BF16_BOUNDS = {
    # absolute / relative deviations of the PRODUCTION path (bf16 autocast, byte planes, every hand-written kernel) from the
    # reference's float32 numbers on fixture G7b; the log-probability bound is absolute (nats), values absolute, scalars relative
    "init": dict(logp=2e-3, values=4e-3, scalars=2e-2, grad_norm=6e-2),
    "sharp": dict(logp=4e-2, values=3e-2, scalars=3e-2, grad_norm=8e-2),
}
# Absolute slack of the scalar losses.  The policy-gradient term of these fixtures is a 1e-2 mean of 64 products of O(1): on the
# sharpened 20 x 20 model (G7c) per-sample log-probability errors of up to 1.6e-2 move it by 1e-3 (measured: 9.6e-4, i.e. 9.5 % of
# 0.0101, with vl and the total loss within 1 %), so the relative bound alone would test the fixture's cancellation, not the kernels.
BF16_ABS_SLACK = {("small", "init"): 2e-4, ("small", "sharp"): 2e-4, ("blox", "init"): 2e-4, ("blox", "sharp"): 2e-3,
                  ("tiny", "init"): 2e-4, ("tiny", "sharp"): 2e-3}
# measured on one MI355X (uint8 and bf16 planes alike):           logp      values    pg rel    vl rel    loss rel  grad norm
#   small init / sharp                                            1.4e-4    1.7e-3    7e-6      7e-5      1e-4      1.4e-4
#                                                                 1.6e-2    1.3e-2    2.7e-2    8.7e-3    1.1e-2    6.3e-3
#   blox  init / sharp                                            1.7e-4    1.4e-3    3.4e-4    1.5e-4    2.8e-4    4e-6
#                                                                 1.6e-2    1.3e-2    9.5e-2    8.1e-3    5.0e-3    6.5e-3


@pytest.mark.parametrize("board", ["small", "blox", "tiny"])
@pytest.mark.parametrize("tag", ["init", "sharp"])
def test_bf16_autocast_loss_tracks_the_reference_fixture(tag, board):
    """The production path -- bf16 autocast on byte planes, fused actor tower, fused feed-forward / projection / LayerNorm
    kernels, hand-written attention, the one-launch loss, paired minibatch -- against numbers the REFERENCE computed in float32
    on its own seeded initialisation (fixtures G7b on smallCapture and G7c on bloxCapture -- the 28-tile tower kernels and the 416-token
    attention; the weights are reproduced from the seed, see test_mappo_cpu).  Bounds are
    stated per quantity in BF16_BOUNDS: per-sample log-probabilities and values absolutely, the scalar losses, the mean entropy
    and the gradient norm relatively.  (BASELINE.json's 1e-4 is the float32 figure, held by the float32 tests.)"""
    from pmx import mappo
    from test_mappo_cpu import _init_batch, reference_init_model
    d, meta, obs, merged, act, old_logp, adv, ret = _init_batch(tag, board)   # board "blox": fixture G7c, 20 x 20 (28-tile tower, 416-token attention)
    bd = BF16_BOUNDS[tag]
    model = reference_init_model(tag, tuple(obs.shape[1:]), meta["seed"]).cuda()
    c = lambda x: x.cuda()
    for in_dtype in (torch.uint8, torch.bfloat16):
        learner = mappo.PPOLearner(model, lr=meta["lr"], autocast_dtype=torch.bfloat16)
        o, m = c(obs).to(in_dtype), c(merged).to(in_dtype)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert model._use_fused_tower(o)
            with learner._shadow_context():
                loss, stats = mappo.ppo_loss(model, o, m, c(act), c(old_logp), c(adv), c(ret), meta["clip_eps"], meta["ent_coef"])
            vals, logp, ent = model.evaluate(o, m, c(act))
        dl = np.abs(logp.detach().float().cpu().numpy() - d[f"{tag}_logp"]).max()
        assert dl <= bd["logp"], ("logp", float(dl))
        dv = np.abs(vals.detach().float().repeat_interleave(2).cpu().numpy() - d[f"{tag}_values"]).max()
        assert dv <= bd["values"], ("values", float(dv))
        for k in ("pg", "vl", "loss"):
            ref = float(d[f"{tag}_{k}"])
            assert abs(float(stats[k]) - ref) <= bd["scalars"] * abs(ref) + BF16_ABS_SLACK[(board, tag)], (k, float(stats[k]), ref)
        ref_e = float(np.mean(d[f"{tag}_entropy"]))
        assert abs(float(stats["entropy"]) - ref_e) <= bd["scalars"] * ref_e
        learner._backward_into_bucket(loss)          # (the heads' gradients arrive through their bf16 shadow copies)
        gn = float(learner.bucket.grad.double().norm())
        assert abs(gn - float(d[f"{tag}_grad_norm"])) <= bd["grad_norm"] * float(d[f"{tag}_grad_norm"]), (gn, float(d[f"{tag}_grad_norm"]))


def test_curriculum_and_pool_rollouts_on_the_gpu():
    """opponent="curriculum" through its three phases and opponent="pool": every mode's rollout (in-kernel randomTeam and
    baselineTeam, self-play, a pool snapshot; the learner on either colour) fills finite buffers, and a full update runs."""
    from pmx import trainer
    tr = trainer.VecMAPPOTrainer("smallCapture", n_envs=128, horizon=6, minibatch=256, epochs=1, seed=11, length=30,
                                 opponent="curriculum", curriculum_scale=0.05, total_updates=100)
    seen, sides = set(), set()
    for idx in [0, 5] + [20] * 40 + [60] * 80:                     # phase thresholds at updates 10 and 40
        if {"random", "baseline", "self", "pool"} <= seen and sides == {True, False}:
            break
        tr.update_idx = idx
        tr.rollout()
        mode, red = tr.stats["opponent"], tr.stats["play_as_red"]
        assert (idx <= 10 and mode == "random") or (10 < idx <= 40 and mode in ("random", "baseline")) or idx > 40
        seen.add(mode)
        if mode in ("self", "pool"):
            sides.add(red)
        for buf in (tr.rew_buf, tr.val_buf, tr.logp_buf):
            assert torch.isfinite(buf).all()
        assert int(tr.act_buf.min()) >= 0 and int(tr.act_buf.max()) <= 4
    assert seen == {"random", "baseline", "self", "pool"} and sides == {True, False}
    tr.compute_gae()
    tr.update()
    assert torch.isfinite(tr.stats["loss"]) and torch.isfinite(tr.stats["grad_norm"])
    tr.env.close()
    tp = trainer.VecMAPPOTrainer("smallCapture", n_envs=64, horizon=4, minibatch=128, epochs=1, seed=12, length=30, opponent="pool")
    for _ in range(3):
        st = tp.train_update()
        assert st["opponent"] == "pool" and torch.isfinite(st["loss"])
    assert len(tp.opponent_pool) == 2                              # the initial snapshot + the one taken at update 0
    tp.env.close()


def test_full_checkpoint_resume_on_the_gpu(tmp_path):
    """save_full / load_full with device tensors and the device generator: the restored state is exact and the generator
    continues with the same draws; the update after a reload follows the original one to within what the LDS float adds of
    the tower's per-channel sums allow (their order is not fixed) (the bit-identical continuation is shown on the CPU, tests/test_mappo_cpu.py)."""
    from pmx import trainer
    mk = lambda: trainer.VecMAPPOTrainer("smallCapture", n_envs=64, horizon=4, minibatch=128, epochs=1, seed=21, length=30,
                                         opponent="random", total_updates=30)
    a = mk()
    a.train_update()
    a.rollout(); a.compute_gae()
    path = str(tmp_path / "full.pt")
    a.save_full(path)
    b = mk()
    b.load_full(path)
    for k in ("obs_buf", "merged_buf", "act_buf", "logp_buf", "adv_buf", "ret_buf"):
        getattr(b, k).copy_(getattr(a, k))
    assert torch.equal(a.learner.bucket.data, b.learner.bucket.data) and torch.equal(a.learner.exp_avg, b.learner.exp_avg)
    assert torch.equal(a.learner.exp_avg_sq, b.learner.exp_avg_sq) and torch.equal(a.learner.ema, b.learner.ema)
    assert torch.equal(a.gen.get_state(), b.gen.get_state()) and a.learner.step_count == b.learner.step_count
    a.update(); b.update()
    diff = (a.learner.bucket.data - b.learner.bucket.data).abs()
    assert float(diff.max()) <= 1e-3 and float(diff.mean()) <= 2e-5, (float(diff.max()), float(diff.mean()))
    assert torch.equal(a.gen.get_state(), b.gen.get_state())
    a.env.close(); b.env.close()


@pytest.mark.parametrize("tokens", [154 * 64, 16 * 3 + 5, 32 * 257])
def test_fused_ffn_layernorm_matches_torch(tokens):
    """pmx_ffn_forward / pmx_ffn_backward (LayerNorm(x + linear2(relu(linear1(x)))), csrc/pmx_critic.hip) against float32 torch
    on the bf16-rounded operands: forward within 2 bf16 ulps of the output scale, input gradient within 2e-2 relative
    (Frobenius), parameter gradients within 2e-2 relative of the float64 CPU reference."""
    from pmx import mappo
    torch.manual_seed(3)
    lin1, lin2, ln = torch.nn.Linear(32, 128).cuda(), torch.nn.Linear(128, 32).cuda(), torch.nn.LayerNorm(32).cuda()
    with torch.no_grad():
        ln.weight.add_(0.2 * torch.randn_like(ln.weight)); ln.bias.add_(0.2 * torch.randn_like(ln.bias))
        lin1.bias.add_(0.3 * torch.randn_like(lin1.bias)); lin2.bias.add_(0.3 * torch.randn_like(lin2.bias))
    x = torch.randn(tokens, 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    dy = (0.1 * torch.randn(tokens, 32, device="cuda")).to(torch.bfloat16)
    params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias, ln.weight, ln.bias]
    y = mappo.ffn_layer_norm(x, lin1, lin2, ln)
    assert y.dtype == torch.bfloat16 and y.shape == x.shape
    got = torch.autograd.grad(y, [x] + params, dy)
    # reference in float64 on the CPU with the operands the kernel sees (bf16 weights, bf16 hidden activations)
    rb = lambda t: t + (t.float().to(torch.bfloat16).to(t.dtype) - t).detach()
    xc = x.detach().cpu().double().requires_grad_(True)
    pc = [p.detach().cpu().double().requires_grad_(True) for p in params]
    h = rb(torch.relu(xc @ rb(pc[0]).T + pc[1]))
    f = h @ rb(pc[2]).T + pc[3]
    yc = torch.nn.functional.layer_norm(xc + f, (32,), pc[4], pc[5], ln.eps)
    want = torch.autograd.grad(yc, [xc] + pc, dy.cpu().double())
    assert (y.detach().cpu().double() - yc.detach()).abs().max().item() <= 2 * 2 ** -8 * max(1.0, yc.abs().max().item())
    for a, b in zip(got, want):
        assert a.shape == b.shape
        rel = ((a.detach().cpu().double() - b).norm() / (b.norm() + 1e-12)).item()
        assert rel < 2e-2, (tuple(a.shape), rel)


@pytest.mark.parametrize("tokens", [154 * 48, 37, 32 * 129 + 16])
def test_fused_in_projection_and_out_projection_layernorm_match_torch(tokens):
    """pmx_tok96_* (qkv = W a + b) and pmx_tok32ln_* (LayerNorm(x + W a + b)) against a float64 CPU reference on the
    bf16-rounded operands: outputs within 2 bf16 ulps of the output scale, all gradients within 2e-2 relative (Frobenius)."""
    from pmx import mappo
    torch.manual_seed(5)
    mha = torch.nn.MultiheadAttention(32, 4).cuda()
    ln = torch.nn.LayerNorm(32).cuda()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.3); mha.out_proj.bias.normal_(0, 0.3)
        ln.weight.add_(0.2 * torch.randn_like(ln.weight)); ln.bias.add_(0.2 * torch.randn_like(ln.bias))
    rb = lambda t: t + (t.float().to(torch.bfloat16).to(t.dtype) - t).detach()
    x = torch.randn(tokens, 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    a = torch.randn(tokens, 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    # in-projection
    dq = (0.1 * torch.randn(tokens, 96, device="cuda")).to(torch.bfloat16)
    qkv = mappo.in_proj96(x, mha)
    assert qkv.shape == (tokens, 96) and qkv.dtype == torch.bfloat16
    got = torch.autograd.grad(qkv, [x, mha.in_proj_weight, mha.in_proj_bias], dq)
    xc = x.detach().cpu().double().requires_grad_(True)
    wc, bc = mha.in_proj_weight.detach().cpu().double().requires_grad_(True), mha.in_proj_bias.detach().cpu().double().requires_grad_(True)
    qc = xc @ rb(wc).T + bc
    want = torch.autograd.grad(qc, [xc, wc, bc], dq.cpu().double())
    assert (qkv.detach().cpu().double() - qc.detach()).abs().max().item() <= 2 * 2 ** -8 * max(1.0, qc.abs().max().item())
    for g_, w_ in zip(got, want):
        assert ((g_.detach().cpu().double() - w_).norm() / (w_.norm() + 1e-12)).item() < 2e-2, tuple(w_.shape)
    # out-projection + residual + LayerNorm
    dy = (0.1 * torch.randn(tokens, 32, device="cuda")).to(torch.bfloat16)
    y = mappo.out_proj_add_layer_norm(x, a, mha.out_proj, ln)
    params = [mha.out_proj.weight, mha.out_proj.bias, ln.weight, ln.bias]
    got = torch.autograd.grad(y, [x, a] + params, dy)
    xc, ac = x.detach().cpu().double().requires_grad_(True), a.detach().cpu().double().requires_grad_(True)
    pc = [p.detach().cpu().double().requires_grad_(True) for p in params]
    yc = torch.nn.functional.layer_norm(xc + ac @ rb(pc[0]).T + pc[1], (32,), pc[2], pc[3], ln.eps)
    want = torch.autograd.grad(yc, [xc, ac] + pc, dy.cpu().double())
    assert (y.detach().cpu().double() - yc.detach()).abs().max().item() <= 2 * 2 ** -8 * max(1.0, yc.abs().max().item())
    for g_, w_ in zip(got, want):
        assert ((g_.detach().cpu().double() - w_).norm() / (w_.norm() + 1e-12)).item() < 2e-2, tuple(w_.shape)


@pytest.mark.parametrize("S,B", [(154, 19), (140, 5), (400, 3)])
def test_batch_major_attention_equals_sequence_major(S, B):
    """pmx_attn8_forward_layout / _backward_layout with batch_major on [B, S, .] tensors give bit for bit what the sequence-major
    kernels give on the transposed tensors: the layout only changes addresses."""
    from pmx import mappo
    torch.manual_seed(S + 7)
    qkv = (torch.randn(S, B, 96, device="cuda") * 1.4).to(torch.bfloat16)
    g = torch.randn(S, B, 32, device="cuda").to(torch.bfloat16)
    a = qkv.clone().requires_grad_(True)
    out_a = mappo.attention8(a)
    out_a.backward(g)
    b = qkv.transpose(0, 1).contiguous().requires_grad_(True)
    out_b = mappo.attention8(b, True)
    out_b.backward(g.transpose(0, 1).contiguous())
    assert torch.equal(out_a.transpose(0, 1), out_b)
    assert torch.equal(a.grad.transpose(0, 1), b.grad)
    with torch.no_grad():
        assert torch.equal(mappo.attention8_forward(b.detach(), batch_major=True), out_b)


def test_batch_major_critic_matches_sequence_major_critic():
    """MAPPOAgent.value under bf16 autocast: the channels-last / batch-major path against the [S, B, E] path of the same
    weights -- same kernels on permuted tokens, so values agree to bf16 rounding of the convolution (MIOpen picks another
    algorithm for NHWC) and the gradients to the same."""
    from pmx import mappo
    torch.manual_seed(3)
    H, W = 11, 14
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    merged = (torch.rand(96, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    res = {}
    m.fused_heads = m.fused_projector = False   # (the head-tail and projector kernels only exist on the batch-major path and have their
                                                #  own tests: here both layouts run the library ops, so that the comparison is about
                                                #  the token layout alone)
    for bm in (False, True):
        m.batch_major_critic = bm
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v = m.value(merged).float()
        (v * torch.linspace(-1, 1, v.numel(), device="cuda")).sum().backward()
        res[bm] = (v.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    m.batch_major_critic = True
    v0, g0 = res[False]
    v1, g1 = res[True]
    assert float((v0 - v1).abs().max()) <= 2e-2 * (float(v0.abs().max()) + 1e-3), float((v0 - v1).abs().max())
    assert g0.keys() == g1.keys() and any(n.startswith("critic_projector") for n in g0)
    for n in g0:
        rel = float((g0[n] - g1[n]).norm() / (g0[n].norm() + 1e-9))
        assert rel <= 5e-2, (n, rel)


@pytest.mark.parametrize("hw", [(11, 14), (20, 20)])
def test_residual_gradient_folded_into_the_in_projection_backward(hw):
    """CriticEncoderLayer.fold_residual_gradient: the layer input's gradient = W_in^T dqkv + (gradient of the residual branch), summed
    inside pmx_tok96_backward_res in float32, against autograd's own bfloat16 add of the two (the switch off).  Same kernels
    otherwise, so values are identical and every gradient agrees to one bfloat16 rounding of the layer-input gradients."""
    from pmx import mappo
    torch.manual_seed(5)
    H, W = hw
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    merged = (torch.rand(70, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    res = {}
    for fold in (False, True):
        for layer in m.critic_transformer.layers:
            layer.fold_residual_gradient = fold
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v = m.value(merged).float()
        (v * torch.linspace(-1, 1, v.numel(), device="cuda")).sum().backward()
        res[fold] = (v.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    v0, g0 = res[False]
    v1, g1 = res[True]
    assert torch.equal(v0, v1)
    assert g0.keys() == g1.keys() and any(n.startswith("critic_projector") for n in g0)
    worst = 0.0
    for n in g0:
        rel = float((g0[n] - g1[n]).norm() / (g0[n].norm() + 1e-12))
        worst = max(worst, rel)
        assert rel <= 1e-2, (n, rel)


def test_tok96_backward_res_adds_the_residual_gradient():
    """pmx_tok96_backward_res against torch: da = dy W + res in float32 rounded once; parameter gradients unchanged by res; res = NULL
    is pmx_tok96_backward."""
    import ctypes as C
    from pmx import _lib, mappo
    lib = _lib.load()
    torch.manual_seed(2)
    T = 1000                                            # not a multiple of 32: ragged last pair
    a = torch.randn(T, 32, device="cuda").to(torch.bfloat16)
    dy = torch.randn(T, 96, device="cuda").to(torch.bfloat16)
    res = torch.randn(T, 32, device="cuda").to(torch.bfloat16)
    w = (torch.randn(96, 32, device="cuda") * 0.2)
    b = torch.randn(96, device="cuda") * 0.1
    pack = mappo.pack_in_proj(w, b)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for tag, r in (("none", None), ("res", res)):
        da = torch.full_like(a, 7.0)
        grad = torch.zeros((1 + _lib.GRAD_PARTIAL_ROWS) * _lib.TOK96_GRAD_FLOATS, dtype=torch.float32, device="cuda")
        _lib.check(lib.pmx_tok96_backward_res(a.data_ptr(), dy.data_ptr(), pack.data_ptr(), None if r is None else r.data_ptr(), da.data_ptr(),
                                              grad.data_ptr(), T, st), "pmx_tok96_backward_res")
        outs[tag] = (da.float(), grad[:_lib.TOK96_GRAD_FLOATS].clone())
    da0 = torch.full_like(a, 7.0)
    grad0 = torch.zeros_like(grad)
    _lib.check(lib.pmx_tok96_backward(a.data_ptr(), dy.data_ptr(), pack.data_ptr(), da0.data_ptr(), grad0.data_ptr(), T, st), "pmx_tok96_backward")
    assert torch.equal(outs["none"][0], da0.float()) and torch.equal(outs["none"][1], grad0[:_lib.TOK96_GRAD_FLOATS])
    assert torch.equal(outs["res"][1], outs["none"][1])                       # the parameter gradients do not see res
    want = dy.float() @ w.to(torch.bfloat16).float() + res.float()
    err = (outs["res"][0] - want).abs().max() / want.abs().max()
    assert float(err) <= 1e-2, float(err)
    # the sum is rounded ONCE: closer to the float32 sum than bf16(bf16(dy W) + res) is allowed to be
    twice = (outs["none"][0].to(torch.bfloat16) + res).float()
    assert float((outs["res"][0] - want).abs().mean()) <= float((twice - want).abs().mean()) + 1e-6
    assert lib.pmx_tok96_backward_res(a.data_ptr(), dy.data_ptr(), pack.data_ptr(), a.data_ptr(), a.data_ptr(), grad.data_ptr(), T, st) != 0   # res == da refused


@pytest.mark.parametrize("B,paired,dtype", [(64, False, torch.float32), (1024, True, torch.float32), (4099 * 2, True, torch.float32),
                                            (1024, True, torch.bfloat16), (3, False, torch.float32)])
def test_fused_ppo_loss_matches_the_torch_objective(B, paired, dtype):
    """pmx_ppo_loss against the torch formulas of ppo_loss (pacman_mappo_resnet.py:571-585) on the same logits / values:
    the five scalars and the gradients with respect to the logits and the values, clipped and unclipped samples alike."""
    from pmx import mappo
    torch.manual_seed(B)
    dev = "cuda"
    logits = (torch.randn(B, 5, device=dev) * 2).to(dtype).requires_grad_(True)
    BV = B // 2 if paired else B
    values = torch.randn(BV, device=dev, requires_grad=True)
    act = torch.randint(0, 5, (B,), device=dev)
    old_logp = torch.log_softmax(torch.randn(B, 5, device=dev), -1).gather(1, act.view(-1, 1)).squeeze(1)
    adv, ret = torch.randn(B, device=dev) * 3 + 0.5, torch.randn(B, device=dev)
    clip_eps, ent_coef = 0.15, 0.02
    stats = mappo._PPOLossFn.apply(logits, values, act, old_logp, adv, ret, clip_eps, ent_coef, mappo.VF_COEF)
    stats[4].backward()
    got = (stats.detach().double(), logits.grad.double(), values.grad.double())
    # the reference formulas in float64 from the same (already rounded) inputs
    z = logits.detach().double().requires_grad_(True)
    v = values.detach().double().requires_grad_(True)
    norm = z - z.logsumexp(-1, keepdim=True)
    probs = torch.softmax(norm, -1)
    logp = norm.gather(1, act.view(-1, 1)).squeeze(1)
    ent = -(norm * probs).sum(-1)
    vv = v.repeat_interleave(2) if paired else v
    a = adv.double()
    na = (a - a.mean()) / (a.std() + 1e-8)
    ratio = (logp - old_logp.double()).exp()
    pg = -torch.min(na * ratio, na * torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps)).mean()
    vl = 0.5 * ((vv - ret.double()) ** 2).mean()
    loss = pg + mappo.VF_COEF * vl - ent_coef * ent.mean()
    loss.backward()
    cf = ((ratio - 1).abs() > clip_eps).double().mean()
    ref = torch.stack([pg, vl, ent.mean(), cf, loss]).detach()
    tol = 2e-5 if dtype == torch.float32 else 2e-5      # the inputs are identical; float32 arithmetic inside the kernel
    assert torch.allclose(got[0], ref, rtol=tol, atol=tol), (got[0], ref)
    gtol = 1e-5 if dtype == torch.float32 else 1e-2     # bfloat16 logits get a bfloat16 gradient
    assert float((got[1] - z.grad).abs().max()) <= gtol * (float(z.grad.abs().max()) + 1e-12), float((got[1] - z.grad).abs().max())
    assert float((got[2] - v.grad).abs().max()) <= 1e-5 * (float(v.grad.abs().max()) + 1e-12)


def test_fused_ppo_loss_is_what_ppo_loss_uses_and_agrees_with_the_torch_path():
    from pmx import mappo
    torch.manual_seed(5)
    H, W = 11, 14
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    B = 64
    obs = (torch.rand(B, 8, H, W, device="cuda") < 0.2).float()
    merged = (torch.rand(B // 2, 8, H, W, device="cuda") < 0.2).float()
    act = torch.randint(0, 5, (B,), device="cuda")
    old_logp, adv, ret = -torch.rand(B, device="cuda") - 1, torch.randn(B, device="cuda"), torch.randn(B, device="cuda")
    out = {}
    for fused in (True, False):
        mappo.MAPPOAgent.fused_loss = fused
        m.zero_grad(set_to_none=True)
        loss, st = mappo.ppo_loss(m, obs, merged, act, old_logp, adv, ret, 0.15, 0.02)
        loss.backward()
        out[fused] = (loss.detach().clone(), {k: v.clone() for k, v in st.items()}, [p.grad.clone() for p in m.parameters() if p.grad is not None])
    mappo.MAPPOAgent.fused_loss = True
    assert abs(float(out[True][0] - out[False][0])) <= 1e-5 * (abs(float(out[False][0])) + 1)
    for k in out[True][1]:
        assert abs(float(out[True][1][k] - out[False][1][k])) <= 1e-5 * (abs(float(out[False][1][k])) + 1), k
    for a, b in zip(out[True][2], out[False][2]):
        assert float((a - b).norm()) <= 2e-4 * (float(b.norm()) + 1e-9)


def test_one_launch_minibatch_gather_equals_torch_indexing():
    """The replayed optimizer step fed by pmx_gather_rows (paired minibatches, reports summed inside the graph) against the
    same replay fed by torch indexing and copies: same random streams, so weights, EMA and the averaged reports agree (not bit
    for bit: the tower's data-gradient kernel adds its per-channel sums with LDS float adds, whose order differs from run to run).
    FOUR optimizer steps, not the whole update: over a whole update (24 steps here) that 1e-7 noise decides, now and then, on which
    side of the clip threshold a borderline sample falls, and two runs of the SAME path then differ by a few 1e-4 in the weights and
    ~0.05 in the averaged gradient norm (tools/r03_flaky.sh) -- a property of PPO's clipping, not of either feeding path."""
    from pmx import trainer
    res = {}
    for gather in (True, False):
        tr = trainer.VecMAPPOTrainer("tinyCapture", 64, horizon=8, minibatch=128, opponent="random", use_graph=True, seed=11)
        tr.graph_gather = gather
        tr.rollout(); tr.compute_gae(); tr.update(max_steps=4)
        res[gather] = (tr.learner.bucket.data.clone(), tr.learner.ema.clone(), {k: float(v) for k, v in tr.stats.items() if k in ("pg", "vl", "entropy", "loss", "grad_norm")})
        tr.env.close()
    for a, b in ((res[True][0], res[False][0]), (res[True][1], res[False][1])):
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()), float((a - b).norm() / b.norm())      # (measured: <= 2e-7)
    for k in res[True][2]:
        assert abs(res[True][2][k] - res[False][2][k]) <= 1e-3 * (abs(res[False][2][k]) + 1e-2), (k, res[True][2][k], res[False][2][k])


def test_gather_rows_matches_index_select():
    import ctypes as C
    from pmx import _lib
    lib = _lib.load()
    torch.manual_seed(2)
    src_a = torch.randint(0, 255, (4096, 8, 7, 20), dtype=torch.uint8, device="cuda")      # 1 120-byte rows
    src_b = torch.randn(4096, device="cuda")
    idx = torch.randperm(2048, device="cuda")[:100]
    dst_a = torch.empty(200, 8, 7, 20, dtype=torch.uint8, device="cuda")
    dst_b = torch.empty(100, device="cuda")
    n = 2
    VP, I32, I64 = C.c_void_p * n, C.c_int32 * n, C.c_int64 * n
    rc = lib.pmx_gather_rows(n, VP(src_a.data_ptr(), src_b.data_ptr()), VP(dst_a.data_ptr(), dst_b.data_ptr()), VP(idx.data_ptr(), idx.data_ptr()),
                             I32(1120, 4), I32(2, 1), I64(200, 100), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    rows = torch.stack((2 * idx, 2 * idx + 1), 1).reshape(-1)
    assert torch.equal(dst_a, src_a[rows]) and torch.equal(dst_b, src_b[idx])
    # the same launch carrying the scalars a replayed graph reads
    dst_a.zero_(); dst_b.zero_()
    sc = torch.full((6,), -1.0, device="cuda")
    rc = lib.pmx_gather_rows_set_floats(n, VP(src_a.data_ptr(), src_b.data_ptr()), VP(dst_a.data_ptr(), dst_b.data_ptr()), VP(idx.data_ptr(), idx.data_ptr()),
                                        I32(1120, 4), I32(2, 1), I64(200, 100), sc.data_ptr(), (C.c_float * 4)(0.25, 1.5, -3.0, 7.0), 4,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.equal(dst_a, src_a[rows]) and torch.equal(dst_b, src_b[idx])
    assert sc.tolist() == [0.25, 1.5, -3.0, 7.0, -1.0, -1.0]


def test_fused_clip_adam_ema_matches_the_torch_ops():
    """pmx_clip_adam_ema against the torch sequence it replaces (clip by the global norm, Adam's single-tensor update, EMA) over
    several steps with gradients on both sides of the clip threshold."""
    from pmx import mappo
    torch.manual_seed(0)
    res = {}
    for fused in (True, False):
        torch.manual_seed(1)
        m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)).cuda()
        L = mappo.PPOLearner(m, lr=3e-4)
        L.fused_optimizer = fused
        norms = []
        for step in range(6):
            g = torch.Generator(device="cuda").manual_seed(100 + step)
            L.bucket.grad.copy_(torch.randn(L.bucket.numel, device="cuda", generator=g) * (0.001 if step % 2 else 0.05))
            if L._use_fused_tail():
                L.step_count += 1
                norms.append(float(L._fused_tail()))
            else:
                gn = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([p.grad for p in L.bucket.params])))
                L.bucket.grad.mul_(torch.clamp(mappo.MAX_GRAD_NORM / (gn + 1e-6), max=1.0))
                L._adam_step()
                L.ema.mul_(mappo.EMA_DECAY).add_(L.bucket.data, alpha=1 - mappo.EMA_DECAY)
                norms.append(float(gn))
        res[fused] = (L.bucket.data.clone(), L.exp_avg.clone(), L.exp_avg_sq.clone(), L.ema.clone(), L.bucket.grad.clone(), norms)
    for a, b, name in zip(res[True][:5], res[False][:5], ("param", "exp_avg", "exp_avg_sq", "ema", "clipped grad")):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), (name, float((a - b).abs().max()))
    assert all(abs(x - y) <= 1e-5 * y for x, y in zip(res[True][5], res[False][5]))


def test_clip_adam_ema_tail_writes_the_bf16_copy_and_the_report_sums():
    """pmx_clip_adam_ema_tail: the same update as pmx_clip_adam_ema (bit for bit), plus the parameters rounded to bfloat16 exactly as
    torch rounds them, plus reports5 / the gradient norm added to the running sums."""
    import ctypes as C
    from pmx import _lib
    lib = _lib.load()
    n = 100_003
    g0 = torch.Generator(device="cuda").manual_seed(5)
    grad0, p0 = torch.randn(n, device="cuda", generator=g0) * 0.01, torch.randn(n, device="cuda", generator=g0)
    m0, v0, e0 = torch.randn(n, device="cuda", generator=g0) * 0.01, torch.rand(n, device="cuda", generator=g0) * 1e-4, p0.clone()
    reports = torch.tensor([0.5, -1.25, 2.0, 0.125, 3.0], device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for tail in (False, True):
        grad, p, m, v, e = (t.clone() for t in (grad0, p0, m0, v0, e0))
        scratch = torch.empty(_lib.OPT_PARTIALS, dtype=torch.float64, device="cuda")
        norm = torch.zeros(1, device="cuda")
        p16 = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        sums = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], device="cuda")
        common = (grad.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), n, scratch.data_ptr(), None, 3e-4, 1.0, 0.9, 0.999,
                  1e-8, 0.5, 0.999, norm.data_ptr())
        if tail:
            rc = lib.pmx_clip_adam_ema_tail(*common, p16.data_ptr(), reports.data_ptr(), sums.data_ptr(), st)
        else:
            rc = lib.pmx_clip_adam_ema(*common, st)
        assert rc == 0
        out[tail] = (grad, p, m, v, e, norm.clone(), p16, sums)
    for a, b in zip(out[True][:6], out[False][:6]):
        assert torch.equal(a, b)
    assert torch.equal(out[True][6], out[True][1].to(torch.bfloat16))
    want = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], device="cuda") + torch.cat([reports, out[True][5]])
    assert torch.allclose(out[True][7], want, rtol=1e-6, atol=0)


@pytest.mark.parametrize("rows", [1, 7, 64, 301])
def test_flatten_sum_adds_partial_rows_while_it_gathers(rows):
    """pmx_flatten_sum_to_f32: tensors that are still partial rows (row 0 of a buffer = where the sum would go, rows 1 .. n behind
    it) arrive summed, ordinary float32 / bfloat16 tensors are copied / widened, all in one launch."""
    import ctypes as C
    from pmx import _lib
    lib = _lib.load()
    torch.manual_seed(rows)
    floats = 1120
    buf = torch.randn((1 + rows) * floats, device="cuda")                      # row 0: garbage the gather must not read
    plain = torch.randn(333, device="cuda")
    half = torch.randn(77, device="cuda").to(torch.bfloat16)
    views = [buf[:1024], buf[1024:1056], plain, buf[1056:1120], half]           # three slices of row 0 among ordinary tensors
    n = len(views)
    offs, o = [], 0
    for t in views:
        offs.append(o); o += t.numel()
    dst = torch.full((o,), float("nan"), device="cuda")
    in_buf = [t.data_ptr() >= buf.data_ptr() and t.data_ptr() < buf.data_ptr() + 4 * floats and t.dtype == torch.float32 for t in views]
    rc = lib.pmx_flatten_sum_to_f32(n, (C.c_void_p * n)(*[t.data_ptr() for t in views]), (C.c_uint8 * n)(*[1 if t.dtype == torch.bfloat16 else 0 for t in views]),
                                    (C.c_int32 * n)(*[rows if f else 0 for f in in_buf]), (C.c_int32 * n)(*[floats if f else 0 for f in in_buf]),
                                    (C.c_int64 * n)(*offs), (C.c_int32 * n)(*[t.numel() for t in views]), dst.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    summed = buf.view(1 + rows, floats)[1:].double().sum(0)
    ref = torch.cat([summed[:1024], summed[1024:1056], plain.double(), summed[1056:1120], half.double()])
    assert float((dst.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(dst[offs[2]:offs[2] + 333], plain) and torch.equal(dst[offs[4]:], half.float())


def test_learner_gradient_with_and_without_the_folded_row_sums_and_the_unit_root():
    """The learner's gradient path (the fused objective differentiated as a 5-vector against the constant unit gradient, second-stage
    row sums folded into the gather) against the plain one (loss.backward() semantics: `torch.autograd.grad(loss, ...)`, a kernel per
    reduction): same gradients up to the order of the float32 row sums and of the tower's LDS adds."""
    from pmx import mappo
    H, W, B = 11, 14, 256
    torch.manual_seed(4)
    obs = (torch.rand(B, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    merged = (torch.rand(B // 2, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    act = torch.randint(0, 5, (B,), device="cuda")
    old_logp, adv, ret = -torch.rand(B, device="cuda") - 1, torch.randn(B, device="cuda"), torch.randn(B, device="cuda")
    torch.manual_seed(9)
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    L = mappo.PPOLearner(m, autocast_dtype=torch.bfloat16)
    grads = {}
    for folded in (True, False):
        L.defer_row_sums = folded
        with L._shadow_context(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            loss, _ = mappo.ppo_loss(m, obs, merged, act, old_logp, adv, ret, 0.2, 0.01)
        if not folded:
            del loss._pmx_stats                          # -> loss_root() hands back the loss itself: the select and the scalings run
        L.bucket.grad.zero_()
        L._backward_into_bucket(loss)
        grads[folded] = L.bucket.grad.clone()
        assert not mappo._PENDING_ROWS
    rel = float((grads[True] - grads[False]).norm() / grads[False].norm())
    assert 0.0 < float(grads[False].norm()) and rel <= 2e-3, rel


def test_bf16_shadow_weights_give_the_autocast_step():
    """The optimizer step with the library-op parameters read from the bfloat16 copy of the bucket (PPOLearner.shadow_weights)
    against the plain autocast step: autocast rounds the same float32 weights to the same bfloat16 values per use, so the
    gradient and the updated weights agree up to the order of the tower's LDS float adds."""
    from pmx import mappo
    H, W, B = 11, 14, 256
    torch.manual_seed(4)
    obs = (torch.rand(B, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    merged = (torch.rand(B // 2, 8, H, W, device="cuda") < 0.2).to(torch.uint8)
    act = torch.randint(0, 5, (B,), device="cuda")
    old_logp, adv, ret = -torch.rand(B, device="cuda") - 1, torch.randn(B, device="cuda"), torch.randn(B, device="cuda")
    res = {}
    for shadow in (True, False):
        torch.manual_seed(9)
        m = mappo.MAPPOAgent((8, H, W)).cuda()
        L = mappo.PPOLearner(m, autocast_dtype=torch.bfloat16)
        L.shadow_weights = shadow
        for _ in range(3):
            st = L.update_minibatch(obs, merged, act, old_logp, adv, ret)
        assert (L._sh16 is not None) == shadow
        res[shadow] = (L.bucket.grad.clone(), L.bucket.data.clone(), float(st["loss"]))
    g_rel = float((res[True][0] - res[False][0]).norm() / res[False][0].norm())
    p_rel = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    assert g_rel <= 2e-2 and p_rel <= 1e-4, (g_rel, p_rel)
    assert abs(res[True][2] - res[False][2]) <= 2e-2 * (abs(res[False][2]) + 1e-2)


@pytest.mark.parametrize("B", [512, 37, 3])
def test_fused_actor_tail_matches_torch(B):
    """pmx_actor_tail_forward / _backward (LayerNorm_512 -> GELU -> Linear_5 of actor_head, pacman_mappo_resnet.py:117-122)
    against torch in float64 on the operands the kernel uses (hidden activations and W2 rounded to bf16): logits within 2e-3
    absolute (bf16 activations of O(1) summed over 512 terms), input and parameter gradients within 2e-2 of their largest entry."""
    from pmx import mappo
    torch.manual_seed(B)
    h = (torch.randn(B, 512, device="cuda") * 1.5 + 0.2).to(torch.bfloat16).requires_grad_(True)
    ln = torch.nn.LayerNorm(512).cuda()
    out = torch.nn.Linear(512, 5).cuda()
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(512, device="cuda")); ln.bias.copy_(0.1 * torch.randn(512, device="cuda"))
        out.weight.mul_(3.0)
    dl = torch.randn(B, 5, device="cuda")
    logits = mappo._ActorTail.apply(h, ln.weight, ln.bias, out.weight, out.bias, ln.eps)
    got = torch.autograd.grad((logits * dl).sum(), [h, ln.weight, ln.bias, out.weight, out.bias])
    bf = lambda t: t.to(torch.bfloat16).double()
    h2 = h.detach().double().requires_grad_(True)
    lw, lb = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    w2 = out.weight.detach().double().requires_grad_(True)
    b2 = out.bias.detach().double().requires_grad_(True)
    g = torch.nn.functional.gelu(torch.nn.functional.layer_norm(h2, (512,), lw, lb, ln.eps))
    g = g + (bf(g.detach()) - g.detach())                                       # straight-through bf16 rounding of the GELU output
    w2r = w2 + (bf(w2.detach()) - w2.detach())
    ref = g @ w2r.t() + b2
    want = torch.autograd.grad((ref * dl.double()).sum(), [h2, lw, lb, w2, b2])
    assert float((logits.double() - ref).abs().max()) <= 2e-3 * (1.0 + float(ref.abs().max()))
    for name, a, b in zip(("dh", "dlnw", "dlnb", "dw2", "db2"), got, want):
        assert a.shape == b.shape
        assert float((a.double() - b).abs().max()) <= 2e-2 * (float(b.abs().max()) + 1e-6), name


@pytest.mark.parametrize("B,S", [(256, 154), (19, 400), (2, 7), (1500, 33)])
def test_fused_critic_tail_matches_torch(B, S):
    """pmx_critic_tail_forward / _backward (mean over tokens -> Linear_512 -> GELU -> Linear_1, pacman_mappo_resnet.py:143-147, :169)
    against torch in float64 with the kernel's bf16 roundings made explicit."""
    from pmx import mappo
    torch.manual_seed(B + S)
    tok = (torch.randn(B, S, 32, device="cuda") * 1.2 + 0.3 * torch.randn(B, 1, 32, device="cuda")).to(torch.bfloat16).requires_grad_(True)
    l1, l2 = torch.nn.Linear(32, 512).cuda(), torch.nn.Linear(512, 1).cuda()
    with torch.no_grad():
        l1.weight.mul_(2.0); l1.bias.add_(0.1 * torch.randn(512, device="cuda")); l2.weight.mul_(4.0)
    dv = torch.randn(B, device="cuda")
    val = mappo._CriticTail.apply(tok, l1.weight, l1.bias, l2.weight, l2.bias)
    got = torch.autograd.grad((val * dv).sum(), [tok, l1.weight, l1.bias, l2.weight, l2.bias])
    ste = lambda t: t + (t.detach().to(torch.bfloat16).double() - t.detach())
    t2 = tok.detach().double().requires_grad_(True)
    ps = [p.detach().double().requires_grad_(True) for p in (l1.weight, l1.bias, l2.weight, l2.bias)]
    pooled = ste(t2.mean(1))
    pre = ste(pooled @ ste(ps[0]).t() + ste(ps[1]))
    ref = (ste(torch.nn.functional.gelu(pre)) @ ste(ps[2]).t()).squeeze(-1) + ps[3]
    want = torch.autograd.grad((ref * dv.double()).sum(), [t2] + ps)
    assert float((val.double() - ref).abs().max()) <= 3e-3 * (1.0 + float(ref.abs().max()))
    for name, a, b in zip(("dtokens", "dw1", "db1", "dw2", "db2"), got, want):
        assert a.shape == b.shape, name
        assert float((a.double() - b).abs().max()) <= 3e-2 * (float(b.abs().max()) + 1e-6), name


def test_fused_heads_are_what_the_model_uses_under_autocast():
    """MAPPOAgent.logits / value take the head-tail kernels under bf16 autocast (byte planes, batch-major critic) and agree with the
    library path (fused_heads off) to bf16 accuracy."""
    from pmx import mappo
    torch.manual_seed(4)
    m = mappo.MAPPOAgent((8, 11, 14), 5, 2).cuda()
    with torch.no_grad():
        m.actor_head[3].weight.mul_(60.0); m.critic_head[2].weight.mul_(5.0)
    obs = (torch.rand(96, 8, 11, 14, device="cuda") < 0.2).to(torch.uint8)
    res = {}
    for on in (True, False):
        m.fused_heads = on
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            res[on] = (m.logits(obs).float(), m.value(obs).float())
    m.fused_heads = True
    assert res[True][0].dtype == torch.float32
    assert float((res[True][0] - res[False][0]).abs().max()) <= 3e-2 * (1.0 + float(res[False][0].abs().max()))
    assert float((res[True][1] - res[False][1]).abs().max()) <= 3e-2 * (1.0 + float(res[False][1].abs().max()))


@pytest.mark.parametrize("layout,B,dtype", [("smallCapture", 300, torch.uint8), ("tinyCapture", 37, torch.bfloat16), ("bloxCapture", 65, torch.uint8),
                                            ("smallCapture", 3, torch.float32), ("bloxCapture", 700, torch.uint8)])
def test_fused_projector_matches_torch(layout, B, dtype):
    """pmx_proj_forward / _backward (critic_projector + positional encoding, pacman_mappo_resnet.py:126-127, :69-95, :164 -> batch-major
    tokens) against torch in float64 with the kernel's roundings made explicit (bf16 weights, the convolution + bias rounded to bf16,
    then the bf16 sum with the bf16 table): tokens within one bf16 ulp of the largest, weight / bias gradients within 1e-2."""
    import pmx
    from pmx import mappo
    lay = pmx.get_layout(layout)
    H, W = lay.height, lay.width
    torch.manual_seed(B)
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    conv = m.critic_projector[0]
    with torch.no_grad():
        conv.bias.add_(0.2 * torch.randn(32, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(B + 1)
    obs = (torch.rand(B, 8, H, W, device="cuda", generator=g) < 0.25).float()
    obs[:, 1] *= torch.randint(1, 6, (B, 1, 1), device="cuda", generator=g).float()
    pe = m._pe_table(H, W, obs.device)
    tok = mappo._Projector.apply(obs.to(dtype), conv.weight, conv.bias, pe)
    assert tok.shape == (B, H * W, 32) and tok.dtype == torch.bfloat16
    dt = torch.randn(B, H * W, 32, device="cuda", generator=g).to(torch.bfloat16)
    dw, db = torch.autograd.grad((tok.float() * dt.float()).sum(), [conv.weight, conv.bias])
    bf = lambda t: t.to(torch.bfloat16).double()
    w64 = conv.weight.detach().double().requires_grad_(True)
    b64 = conv.bias.detach().double().requires_grad_(True)
    wr = w64 + (bf(w64.detach()) - w64.detach())
    y = torch.nn.functional.conv2d(obs.double(), wr, b64, padding=1)                     # [B, 32, H, W]
    y = y + (bf(y.detach()) - y.detach())
    ref = (y.permute(0, 2, 3, 1).reshape(B, H * W, 32) + bf(pe)[None])
    ref_r = bf(ref.detach())
    scale = float(ref_r.abs().max())
    assert float((tok.double() - ref_r).abs().max()) <= scale * 2 ** -7, float((tok.double() - ref_r).abs().max())
    dw_ref, db_ref = torch.autograd.grad((ref * dt.double()).sum(), [w64, b64])
    assert float((dw.double() - dw_ref).abs().max()) <= 1e-2 * float(dw_ref.abs().max())
    assert float((db.double() - db_ref).abs().max()) <= 1e-2 * float(db_ref.abs().max())


def test_fused_projector_is_what_the_model_uses_and_agrees_with_the_library_path():
    from pmx import mappo
    torch.manual_seed(9)
    m = mappo.MAPPOAgent((8, 11, 14), 5, 2).cuda()
    with torch.no_grad():
        m.critic_head[2].weight.mul_(5.0)
    obs = (torch.rand(64, 8, 11, 14, device="cuda") < 0.2).to(torch.uint8)
    res = {}
    for on in (True, False):
        m.fused_projector = on
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v = m.value(obs).float()
        v.sum().backward()
        res[on] = (v.detach().clone(), m.critic_projector[0].weight.grad.clone(), m.critic_projector[0].bias.grad.clone())
    m.fused_projector = True
    assert float((res[True][0] - res[False][0]).abs().max()) <= 3e-2 * (1.0 + float(res[False][0].abs().max()))
    for a, b in zip(res[True][1:], res[False][1:]):
        assert float((a - b).norm() / (b.norm() + 1e-9)) <= 6e-2
