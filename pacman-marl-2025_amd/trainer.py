"""Vectorised MAPPO training loop on the GPU env (the rollout / GAE / PPO-update part of pacman_mappo_resnet.train()).

Per update (pacman_mappo_resnet.py:385-600), with N envs in lock-step instead of one:
  rollout  T ticks: canonicalise (red learners) and merge the two learner observations, sample both learners'
           actions from the actor and one value from the critic, step the env, add the heuristic shaping
           (:241-264, :513-522), store everything in DEVICE-resident buffers (:449-455 kept them on the CPU)
  GAE      pmx_gae over the [T][2N] series (:548-553)
  update   UPDATE_EPOCHS x minibatches of the flattened [T*N*2] samples through PPOLearner (:556-595)
Opponents: the in-kernel randomTeam and baselineTeam bots (PMX_ACTION_RANDOM_LEGAL / PMX_ACTION_BASELINE_*), the current
network, or a frozen EMA snapshot from the opponent pool (:396-438; the A*/MCTS/approx-Q teams of the reference's
curriculum are out of scope).
"""
import copy
import ctypes as C
from collections import deque

import numpy as np
import torch

from . import _lib, mappo
from .vec_env import PmxVecEnv

UPDATE_EPOCHS = 3                 # pacman_mappo_resnet.py:24
OPPONENT_POOL_SIZE = 80           # :32
OPPONENT_UPDATE_FREQ = 15         # :33


def shaping_from_agent_words(prev, cur):
    """compute_heuristic_shaping (pacman_mappo_resnet.py:251-264) from the compact self plane.
    prev/cur: int32 [...]: x | y << 8 | carry << 16 of the same agent in two consecutive observations.  float64."""
    px, py, pc = prev & 0xFF, (prev >> 8) & 0xFF, (prev >> 16) & 0xFFFF
    cx, cy = cur & 0xFF, (cur >> 8) & 0xFF
    moved = (px - cx).abs() + (py - cy).abs()
    living = -0.15
    died = (living - 0.4) - (0.15 * pc.to(torch.float64))
    out = torch.full(prev.shape, living, dtype=torch.float64, device=prev.device)
    out = torch.where(moved == 0, torch.full_like(out, living - 0.05), out)     # dist_moved < 0.1
    out = torch.where(moved > 1, died, out)                                     # dist_moved > 1.5
    return out


_OBS_CODES = {torch.float32: _lib.OBS_F32, torch.bfloat16: _lib.OBS_BF16, torch.uint8: _lib.OBS_U8}


def _plane_kernel_ok(*ts):
    return all(t.is_cuda and t.dim() == 4 and t.shape[1] == 8 and t.dtype in _OBS_CODES and t.dtype == ts[0].dtype for t in ts)


def merge_obs(a, b):
    """merge_obs_for_critic (pacman_mappo_resnet.py:267-274) on batches [N,8,H,W]: pmx_merge_obs on device tensors of an
    observation dtype, the torch formulation otherwise."""
    if _plane_kernel_ok(a, b) and a.shape == b.shape:
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        st = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        _lib.check(_lib.load().pmx_merge_obs(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[2], a.shape[3],
                                             _OBS_CODES[a.dtype], st), "pmx_merge_obs")
        return out
    m = a.clone()
    m[:, 1] = torch.maximum(torch.maximum(a[:, 1], b[:, 1]), torch.zeros_like(a[:, 1]))
    m[:, 4] = 0
    return m


def canonicalize_obs(o):
    """canonicalize_obs for a red learner (pacman_mappo_resnet.py:215-229): flip x, swap planes 2<->3, 6<->7; pmx_canonicalize_obs
    on device tensors [N,8,H,W] of an observation dtype, the torch formulation otherwise."""
    if _plane_kernel_ok(o):
        o = o.contiguous()
        out = torch.empty_like(o)
        st = C.c_void_p(torch.cuda.current_stream(o.device).cuda_stream)
        _lib.check(_lib.load().pmx_canonicalize_obs(o.data_ptr(), out.data_ptr(), o.shape[0], o.shape[2], o.shape[3],
                                                    _OBS_CODES[o.dtype], st), "pmx_canonicalize_obs")
        return out
    return torch.flip(o, dims=[-1])[..., [0, 1, 3, 2, 4, 5, 7, 6], :, :]


class VecMAPPOTrainer:
    def __init__(self, layout, n_envs, horizon=32, minibatch=512, epochs=UPDATE_EPOCHS, obs_dtype=None,
                 device="cuda:0", seed=0, rank=0, world_size=1, process_group=None, total_updates=2000, length=300,
                 use_autocast=True, opponent="random", use_graph=False, algorithm="mappo", paired_minibatches=True, flat_bf16=False, curriculum_scale=1.0,
                 env=None, redraw_layouts=False, force_collectives=False):
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        if obs_dtype is None:
            # byte planes on the bf16 path: the values are small integers, the fused actor tower reads bytes directly and
            # the rollout buffers are half the size; float32 planes (the reference's dtype) on the float32 path
            obs_dtype = "uint8" if use_autocast else "float32"
        # `env`: an already built vectorised env (anything with PmxVecEnv's surface).  The host-side tests pass a stand-in to
        # drive update() / the opponent schedule without a GPU; the product always builds the HIP env here, which raises
        # without a GPU
        self.env = env if env is not None else PmxVecEnv(
            layout, n_envs, length=length, reward_forLegalAction=True, defenceReward=True, auto_reset=True, obs_dtype=obs_dtype,
            device=self.device, seed=seed * 1000003 + rank, bots=opponent in ("baseline", "curriculum"), redraw_layouts=redraw_layouts)
        self.N, self.T = n_envs, horizon
        self.minibatch, self.epochs = minibatch, epochs
        # "mappo": centralised critic on merge_obs_for_critic of the two learners (the reference).  "ippo": the same network
        # with the critic fed each agent's OWN observation (BASELINE config 5 names the comparison; the reference has no
        # IPPO, so this variant has no parity target -- SURVEY section 0)
        assert algorithm in ("mappo", "ippo")
        self.algorithm = algorithm
        # the reference's curriculum switches phases at updates 200 and 800 of 2 048 ticks each; a vectorised update holds
        # n_envs * horizon ticks, so the thresholds can be compressed (curriculum_scale 0.1 -> updates 20 and 80)
        self.curriculum_scale = float(curriculum_scale)
        # paired_minibatches: a minibatch is drawn as (env-tick) PAIRS, both learners of a pair together, so that the
        # centralised critic runs once per pair instead of once per agent sample (its input is the same merged observation
        # for both).  Every epoch is still a random permutation that visits each sample once and the loss of a minibatch is
        # the reference's; only the composition of the minibatches differs from the reference's independent shuffle of
        # agent samples (pacman_mappo_resnet.py:566-569).  False restores that shuffle.
        self.paired = bool(paired_minibatches) and algorithm == "mappo" and minibatch % 2 == 0
        self.graph_gather = True     # replayed steps assemble their minibatch with pmx_gather_rows (False: torch indexing + copies)
        self.total_updates = total_updates
        H, W = self.env.layout.height, self.env.layout.width
        self.obs_shape = (8, H, W)
        torch.manual_seed(seed)                                   # identical initial weights on every rank
        self.model = mappo.MAPPOAgent(self.obs_shape, 5, 2).to(self.device)
        self.autocast_dtype = torch.bfloat16 if use_autocast else None
        # force_collectives: issue the gradient all-reduce on a one-rank group too (bench.py rehearses the collective path on a
        # one-GPU box with it)
        self.learner = mappo.PPOLearner(self.model, process_group=process_group, world_size=world_size,
                                        autocast_dtype=self.autocast_dtype, force_collectives=force_collectives)
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.learner.bucket.data, src=0, group=process_group)
            self.learner.ema.copy_(self.learner.bucket.data)
        if flat_bf16:       # optimizer step on one flat bfloat16 weight copy instead of autocast (PPOLearner.enable_bf16_flat)
            if not use_autocast:
                raise ValueError("flat_bf16 is the bfloat16 training path; it replaces autocast in the optimizer step")
            self.learner.enable_bf16_flat()
        self.opponent_model = mappo.MAPPOAgent(self.obs_shape, 5, 2).to(self.device)
        self.opponent_model.load_state_dict(self.model.state_dict())
        self.opponent_model.eval()
        self.opponent_pool = deque(maxlen=OPPONENT_POOL_SIZE)
        self.opponent_pool.append(self.learner.ema_state_dict())
        self.opponent_mode = opponent                              # "random" | "baseline" | "self" | "pool" | "curriculum"
        self.gen = torch.Generator(device=self.device).manual_seed(seed * 7919 + rank)   # action sampling, minibatch order: per rank
        # opponent mode / side / pool draws: the SAME stream on every rank, so that all ranks run the same kind of rollout
        # (an extra opponent forward per tick on some ranks only would make the others wait at every gradient all-reduce)
        self.np_rng = np.random.RandomState(seed * 31)
        dt, dev, N, T = self.env.obs_torch_dtype, self.device, n_envs, horizon
        self.obs_buf = torch.zeros((T, N, 2) + self.obs_shape, dtype=dt, device=dev)
        self.merged_buf = torch.zeros((T, N) + self.obs_shape, dtype=dt, device=dev)
        self.act_buf = torch.zeros((T, N, 2), dtype=torch.int64, device=dev)
        self.logp_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        self.rew_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        self.done_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        self.val_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        self.adv_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        self.ret_buf = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        starts = self.env.layout.starts.astype(np.int64)
        self.start_words = torch.tensor([int(starts[i, 0]) | (int(starts[i, 1]) << 8) for i in range(4)],
                                        dtype=torch.int32, device=dev)
        self.env.reset()
        self._acts = None                                          # persistent per-rollout tensors, allocated on first use
        self.prev_agent = self.start_words[None].expand(N, 4).clone()
        self.update_idx = 0
        self.stats = {}
        # Optional: replay the optimizer step from a hipGraph when every minibatch has the same shape (launch-bound at the
        # reference's 512 samples: 1.4x more optimizer steps/s).  Needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (set by the package
        # on import; ROCm 7's graph packet capture corrupts a node after ~230 interleaved replays); the
        # update loop also checks the gradient norm once per update so that a non-finite policy is never sampled from.
        # (data parallel: the replayed step is one graph per gradient group with the eager RCCL all-reduce of that group between
        # them -- PPOLearner.capture -- so the collectives are the ones the eager step issues)
        if use_graph and not mappo.PPOLearner.graph_replay_safe():
            raise ValueError("use_graph needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 exported before HIP initialises (DESIGN.md section 5)")
        self.use_graph = bool(use_graph) and (horizon * n_envs * 2) % minibatch == 0
        self._graph_ready = False

    # ---------------------------------------------------------------------------------------------------------
    def _net_in(self, x):
        """Observation tensors are small integers: under autocast the network takes bytes or bf16 as they are (the fused
        actor tower reads either, the critic converts on entry), fp32 otherwise."""
        if self.autocast_dtype is not None:
            return x if x.dtype in (torch.uint8, torch.bfloat16) else x.to(torch.bfloat16)
        return x.float()

    def _freeze_tower_packs(self, on):
        """The rollout runs thousands of inference calls on frozen weights: pack the actor towers' parameters once."""
        from . import actor_tower
        H, W = self.obs_shape[1], self.obs_shape[2]
        for m in (self.model, self.opponent_model):
            m.tower_pack = None
            if on and self.autocast_dtype is not None and m.fused_tower and actor_tower.tower_supported(H, W):
                m.tower_pack = actor_tower.pack_params(actor_tower._tower_params(m.actor_backbone))

    def _forward_policy(self, model, obs2, merged, want_value):
        ctx = torch.autocast(device_type=self.device.type, dtype=self.autocast_dtype) if self.autocast_dtype else _NullCtx()
        with torch.no_grad(), ctx:
            a, lp = model.act(self._net_in(obs2.reshape((-1,) + self.obs_shape)), generator=self.gen)
            v = model.value(self._net_in(merged)).float() if want_value else None
        return a.view(-1, 2), lp.view(-1, 2), v

    def _pick_opponent(self):
        """The self-play part of the curriculum (pacman_mappo_resnet.py:396-438) with the opponents this build has."""
        mode = self.opponent_mode
        if mode == "curriculum":
            # :396-438 with the opponents this build has on the GPU: randomTeam first, then randomTeam / baselineTeam
            # (the reference weights its "hard" teams, baselineTeam among them, 5x), then 40 % self / 20 % pool / 40 % bots
            if self.update_idx <= 200 * self.curriculum_scale:
                mode = "random"
            elif self.update_idx <= 800 * self.curriculum_scale:
                mode = "baseline" if self.np_rng.rand() < 5.0 / 6.0 else "random"
            else:
                r = self.np_rng.rand()
                mode = "self" if r < 0.40 else ("pool" if r < 0.60 else ("baseline" if self.np_rng.rand() < 5.0 / 6.0 else "random"))
        play_as_red = False
        if mode == "self":
            play_as_red = bool(self.np_rng.rand() > 0.5)
            self.opponent_model.load_state_dict(self.model.state_dict())
        elif mode == "pool":
            play_as_red = bool(self.np_rng.rand() > 0.5)
            self.opponent_model.load_state_dict(self.opponent_pool[self.np_rng.randint(len(self.opponent_pool))])
        return mode, play_as_red

    def rollout(self):
        """T ticks.  Per tick (pacman_mappo_resnet.py:461-546, N envs at once): the env itself writes the learners' (canonicalised)
        observations into obs_buf[t] and the merged critic input into merged_buf[t] (pmx_emit_team_obs -- no select / flip /
        copy / merge passes over the planes, and the four-agent observation tensor of pmx_step is not produced at all), the
        policy samples both learners' actions and one value, the env steps with the actions taken from a persistent tensor."""
        env, N, T = self.env, self.N, self.T
        mode, red = self._pick_opponent()
        self._freeze_tower_packs(True)
        learner_ids = [0, 2] if red else [1, 3]
        opp_ids = [1, 3] if red else [0, 2]
        team = 0 if red else 1
        ep_ret = torch.zeros((), dtype=torch.float64, device=self.device)
        n_done = torch.zeros((), dtype=torch.int64, device=self.device)
        n_win = torch.zeros((), dtype=torch.int64, device=self.device)
        if self._acts is None:
            dt = self.obs_buf.dtype
            self._acts = torch.empty((N, 4), dtype=torch.int8, device=self.device)
            self._opp_obs = torch.empty((N, 2) + self.obs_shape, dtype=dt, device=self.device)
            self._boot_obs = torch.empty((N, 2) + self.obs_shape, dtype=dt, device=self.device)
            self._boot_merged = torch.empty((N,) + self.obs_shape, dtype=dt, device=self.device)
        acts = self._acts
        acts.fill_(_lib.ACTION_RANDOM_LEGAL)
        if mode == "baseline":                                       # createTeam: first index offensive, second defensive
            acts[:, opp_ids[0]] = _lib.ACTION_BASELINE_OFFENSE
            acts[:, opp_ids[1]] = _lib.ACTION_BASELINE_DEFENSE
        mappo_alg = self.algorithm == "mappo"
        for t in range(T):
            env.emit_team_obs(red, self.obs_buf[t], self.merged_buf[t] if mappo_alg else None)
            if mappo_alg:
                a, lp, v = self._forward_policy(self.model, self.obs_buf[t], self.merged_buf[t], True)
                self.val_buf[t].copy_(v[:, None].expand(N, 2))
            else:
                a, lp, v = self._forward_policy(self.model, self.obs_buf[t], self.obs_buf[t].view((-1,) + self.obs_shape), True)
                self.val_buf[t].copy_(v.view(N, 2))
            self.act_buf[t].copy_(a)
            self.logp_buf[t].copy_(lp)
            acts[:, learner_ids] = mappo.canonicalize_action(a, red).to(torch.int8)
            if mode in ("self", "pool"):
                env.emit_team_obs(not red, self._opp_obs, None)      # the opponent is red when the learner is blue
                oa, _, _ = self._forward_policy(self.opponent_model, self._opp_obs, None, False)
                acts[:, opp_ids] = mappo.canonicalize_action(oa, not red).to(torch.int8)
            _, rew, done, info = env.step(acts, want_obs=False)
            cur_agent = info["agent"]
            shp = shaping_from_agent_words(self.prev_agent[:, learner_ids], cur_agent[:, learner_ids])   # [N,2] f64
            team_reward = rew[:, team] + rew[:, team]                # sum over the two learners' (identical) rewards
            self.rew_buf[t].copy_((team_reward[:, None] + mappo.SHAPING_SCALE * shp).to(torch.float32))
            d = done.to(torch.bool)
            self.done_buf[t].copy_(done.to(torch.float32)[:, None].expand(N, 2))
            self.prev_agent = torch.where(d[:, None], self.start_words[None].expand(N, 4), cur_agent)
            ep_ret += team_reward.sum()
            n_done += d.sum()
            sc = info["score"]
            n_win += (d & ((sc > 0) if red else (sc < 0))).sum()
        # bootstrap value of the state after the last tick (:541-546)
        env.emit_team_obs(red, self._boot_obs, self._boot_merged if mappo_alg else None)
        ctx = torch.autocast(device_type=self.device.type, dtype=self.autocast_dtype) if self.autocast_dtype else _NullCtx()
        with torch.no_grad(), ctx:
            if mappo_alg:
                self.last_value = self.model.value(self._net_in(self._boot_merged)).float()[:, None].expand(N, 2).contiguous()
            else:
                self.last_value = self.model.value(self._net_in(self._boot_obs.view((-1,) + self.obs_shape))).float().view(N, 2)
        self._freeze_tower_packs(False)
        self.stats.update(opponent=mode, play_as_red=red, rollout_reward=ep_ret, episodes=n_done, wins=n_win)

    def compute_gae(self):
        T, n = self.T, self.N * 2
        last = self.last_value.contiguous()                         # [N, 2]
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.env.lib.pmx_gae(self.rew_buf.data_ptr(), self.val_buf.data_ptr(), self.done_buf.data_ptr(),
                                        last.data_ptr(), T, n, mappo.GAMMA, mappo.GAE_LAMBDA, self.adv_buf.data_ptr(),
                                        self.ret_buf.data_ptr(), st), "pmx_gae")

    def update(self, max_steps=None):
        """The PPO epochs over the rollout buffers.  max_steps stops after that many optimizer steps (warm-up runs only)."""
        lr, ent_coef, clip_eps = mappo.schedule(self.update_idx, self.total_updates)
        self.learner.set_lr(lr)
        S = self.T * self.N * 2
        obs = self.obs_buf.view((S,) + self.obs_shape)
        merged = self.merged_buf.view((self.T * self.N,) + self.obs_shape)
        act, logp = self.act_buf.view(S), self.logp_buf.view(S)
        adv, ret = self.adv_buf.view(S), self.ret_buf.view(S)
        agg = None
        steps = 0
        snap = None
        # the one-launch minibatch gather needs the rollout tensors in the very type the graph's inputs have
        gather = bool(self.use_graph and self.graph_gather and self.paired and self.device.type == "cuda" and S % self.minibatch == 0
                      and self._net_in(obs[:1]).dtype == obs.dtype and self._net_in(merged[:1]).dtype == merged.dtype)
        if self.use_graph:      # a bad replay must not leave NaNs in the weights, the Adam moments and the EMA: keep a copy to go back to
            L = self.learner
            snap = [t.clone() for t in (L.bucket.data, L.exp_avg, L.exp_avg_sq, L.ema)] + [L.step_count]
        for _ in range(self.epochs):
            if self.paired:
                pperm = torch.randperm(S // 2, device=self.device, generator=self.gen)
            else:
                perm = torch.randperm(S, device=self.device, generator=self.gen)
            for s0 in range(0, S, self.minibatch):
                if self.paired:
                    pr = pperm[s0 // 2:(s0 + self.minibatch) // 2]
                else:
                    mb = perm[s0:s0 + self.minibatch]
                if self.use_graph and not self._graph_ready:
                    self.learner.capture(self.minibatch, self.obs_shape, self._net_in(obs[:1]).dtype,
                                         clip_eps, ent_coef, merged_batch=self.minibatch // 2 if self.paired else None)
                    self._graph_ready = True
                if gather:
                    # replayed step, paired minibatch: one gather launch into the graph's inputs, reports summed inside the graph
                    if steps == 0:
                        self.learner._g_acc.zero_()
                    self.learner.update_minibatch_graph_gather(
                        {"obs": obs, "merged": merged, "act": act, "logp": logp, "adv": adv, "ret": ret}, pr,
                        {"obs": 2, "merged": 1, "act": 2, "logp": 2, "adv": 2, "ret": 2}, clip_eps, ent_coef)
                    steps += 1
                    if max_steps is not None and steps >= max_steps:
                        break
                    continue
                mb = torch.stack((2 * pr, 2 * pr + 1), dim=1).reshape(-1) if self.paired else mb   # rows 2k, 2k+1 = the two learners of pair k
                step = self.learner.update_minibatch_graph if self.use_graph else self.learner.update_minibatch
                critic_in = merged[pr] if self.paired else (merged[mb // 2] if self.algorithm == "mappo" else obs[mb])
                st = step(self._net_in(obs[mb]), self._net_in(critic_in), act[mb], logp[mb], adv[mb], ret[mb], clip_eps, ent_coef)
                steps += 1
                agg = {k: v.clone() for k, v in st.items()} if agg is None else {k: agg[k] + st[k] for k in agg}
                if max_steps is not None and steps >= max_steps:
                    break
            if max_steps is not None and steps >= max_steps:
                break
        if gather:
            acc = self.learner._g_acc / steps
            agg = {k: acc[i] * steps for i, k in enumerate(self.learner._g_acc_keys)}
        self.stats.update({k: v / steps for k, v in agg.items()})
        if self.use_graph and not bool(torch.isfinite(self.stats["grad_norm"]).item()):
            L = self.learner
            for t, v in zip((L.bucket.data, L.exp_avg, L.exp_avg_sq, L.ema), snap[:4]):
                t.copy_(v)
            L.step_count = snap[4]
            L._refresh_bf16()
            raise RuntimeError("non-finite gradient norm from the graph-replayed optimizer step; the learner was restored to "
                               "its state before this update")
        self.stats.update(lr=lr, ent_coef=ent_coef, clip_eps=clip_eps, optimizer_steps=steps)
        if self.update_idx % OPPONENT_UPDATE_FREQ == 0:
            self.opponent_pool.append(self.learner.ema_state_dict())
        self.update_idx += 1

    def train_update(self):
        self.rollout()
        self.compute_gae()
        self.update()
        return self.stats

    def save_ema(self, path):
        """The reference's checkpoint: EMA weights only (pacman_mappo_resnet.py:647-651)."""
        torch.save(self.learner.ema_state_dict(), path)

    def save_full(self, path):
        """Full resume state, which the reference lacks: weights, EMA, Adam moments, opponent pool,
        counters and both random streams -- tensors, numbers and strings only, so that it loads with weights_only=True.
        The games themselves are not saved: a resumed run starts its envs from fresh episodes."""
        kind, keys, pos, has_gauss, cached = self.np_rng.get_state()
        torch.save({"data": self.learner.bucket.data, "ema": self.learner.ema, "exp_avg": self.learner.exp_avg,
                    "exp_avg_sq": self.learner.exp_avg_sq, "step": int(self.learner.step_count), "update": int(self.update_idx),
                    "total_updates": int(self.total_updates), "pool": list(self.opponent_pool), "gen": self.gen.get_state(),
                    "np_rng": {"kind": str(kind), "keys": torch.from_numpy(np.asarray(keys, dtype=np.int64)), "pos": int(pos),
                               "has_gauss": int(has_gauss), "cached_gaussian": float(cached)}}, path)

    def load_full(self, path):
        ck = torch.load(path, map_location=self.device, weights_only=True)
        need = ("data", "ema", "exp_avg", "exp_avg_sq", "step", "update", "total_updates", "pool", "gen", "np_rng")
        if not isinstance(ck, dict) or any(k not in ck for k in need) or not isinstance(ck["np_rng"], dict) or "keys" not in ck["np_rng"]:
            raise ValueError(f"{path}: not a full-resume checkpoint of this version (save_full writes the keys {', '.join(need)}; "
                             "checkpoints written before `total_updates` and the tensor-valued `np_rng` were added are not supported)")
        if int(ck["total_updates"]) != int(self.total_updates):
            raise ValueError(f"checkpoint was written for a schedule of {ck['total_updates']} updates, this trainer has {self.total_updates}")
        self.learner.bucket.data.copy_(ck["data"]); self.learner.ema.copy_(ck["ema"])
        self.learner.exp_avg.copy_(ck["exp_avg"]); self.learner.exp_avg_sq.copy_(ck["exp_avg_sq"])
        self.learner.step_count, self.update_idx = int(ck["step"]), int(ck["update"])
        self.opponent_pool = deque(ck["pool"], maxlen=OPPONENT_POOL_SIZE)
        self.gen.set_state(ck["gen"].cpu())
        r = ck["np_rng"]
        self.np_rng.set_state((r["kind"], r["keys"].cpu().numpy().astype(np.uint32), int(r["pos"]), int(r["has_gauss"]),
                               float(r["cached_gaussian"])))
        self.learner._refresh_bf16()


def evaluate_vs_bots(model, num_episodes=20, layout_file="bloxCapture", teams=("baselineTeam", "randomTeam"), length=300,
                     device="cuda:0"):
    """evaluate_vs_bots (pacman_mappo_resnet.py:293-337): a fresh env per episode against host CaptureAgent bots (red),
    the learner plays blue with argmax actions; returns (mean return, std, win rate), a win being final score < 0.
    The reference cycles through its A*/approx-Q/baseline/MCTS teams; here through the team files this build ships."""
    from .gym_env import gymPacMan_parallel_env
    was_training = model.training
    model.eval()
    returns, wins = [], 0
    learner_ids = [1, 3]
    for ep in range(num_episodes):
        env = gymPacMan_parallel_env(layout_file=layout_file, display=False, reward_forLegalAction=True, defenceReward=True,
                                     length=length, enemieName=teams[ep % len(teams)], self_play=False, device=device)
        obs_dict, _ = env.reset()
        ep_ret, done = 0.0, False
        while not done:
            lo = torch.stack([obs_dict[env.agents[i]].float() for i in learner_ids])
            with torch.no_grad():
                acts = model.get_deterministic_action(lo.to(next(model.parameters()).device)).cpu()
            obs_dict, rewards, dones, _ = env.step({env.agents[a]: int(acts[k]) for k, a in enumerate(learner_ids)})
            ep_ret += sum(rewards[env.agents[i]] for i in learner_ids)
            done = any(dones.values())
        returns.append(ep_ret)
        wins += int(env.game.state.data.score < 0)
        env.close()
    if was_training:
        model.train()
    return float(np.mean(returns)), float(np.std(returns)), wins / num_episodes


@torch.no_grad()
def evaluate_vectorized(model, layout="bloxCapture", n_envs=1024, opponent="baseline", length=300, device="cuda:0", seed=0,
                        autocast=True):
    """evaluate_vs_bots (pacman_mappo_resnet.py:293-337) for n_envs episodes at once: the learner plays blue with argmax
    actions against the in-kernel randomTeam / baselineTeam, one episode per env (an env stops counting at its first done).
    Returns (mean episode return of the learner team as the reference sums it, std, win rate = share with final score < 0)."""
    dev = torch.device(device)
    env = PmxVecEnv(layout, n_envs, length=length, auto_reset=True, obs_dtype="bfloat16" if autocast else "float32", device=dev,
                    seed=seed, bots=(opponent == "baseline"))
    obs, _ = env.reset()
    was_training = model.training
    model.eval()
    alive = torch.ones(n_envs, dtype=torch.bool, device=dev)
    ret = torch.zeros(n_envs, dtype=torch.float64, device=dev)
    final = torch.zeros(n_envs, dtype=torch.int32, device=dev)
    opp = (_lib.ACTION_BASELINE_OFFENSE, _lib.ACTION_BASELINE_DEFENSE) if opponent == "baseline" else (_lib.ACTION_RANDOM_LEGAL,) * 2
    ctx = torch.autocast(device_type=dev.type, dtype=torch.bfloat16) if autocast else _NullCtx()
    for _ in range(length + 1):
        lo = obs[:, [1, 3]].reshape((-1,) + tuple(obs.shape[2:]))
        with ctx:
            a = model.get_deterministic_action(lo.to(torch.bfloat16) if autocast else lo.float()).view(n_envs, 2)
        acts = torch.empty((n_envs, 4), dtype=torch.int8, device=dev)
        acts[:, 0], acts[:, 2] = opp[0], opp[1]
        acts[:, [1, 3]] = a.to(torch.int8)
        obs, rew, done, info = env.step(acts)
        ret += torch.where(alive, rew[:, 1] + rew[:, 1], torch.zeros_like(ret))     # sum over both learners' rewards (:328)
        d = done.to(torch.bool) & alive
        final = torch.where(d, info["score"], final)
        alive &= ~d
        if not bool(alive.any()):
            break
    env.close()
    if was_training:
        model.train()
    return float(ret.mean()), float(ret.std()), float((final < 0).double().mean())


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
