"""GPU tests of the drop-in boundary: gymPacMan_parallel_env (dict API) against the golden trajectories, and the
CaptureAgent bots against G9 -- the red team's action strings, scores, rewards and dones recorded from the reference
under random.seed(k) (stream-exact: same stdlib `random` call order)."""
import random

import numpy as np
import pytest
import torch

import _golden as G

pytestmark = pytest.mark.gpu


def _write_layout(tmp_path, rows, name):
    p = tmp_path / name
    p.write_text("\n".join(rows))
    return str(p)


@pytest.mark.parametrize("name", ["traj_small_hunter.npz", "traj_maze23_hunter.npz"])
def test_dict_api_self_play_matches_reference(tmp_path, name):
    import pmx
    d, meta = G.load(name)
    path = _write_layout(tmp_path, meta["layout"], "l.lay")
    env = pmx.gymPacMan_parallel_env(layout_file=path, length=meta["length"], self_play=True)
    assert env.agents == [0, 1, 2, 3]
    obs, info = env.reset()
    assert tuple(env.get_Observation(0).shape) == tuple(d["init_obs"].shape[1:])
    for i in range(4):
        assert obs[i].dtype == torch.float32 and (obs[i].cpu().numpy() == d["init_obs"][i]).all()
    T = 330
    for t in range(T):
        o, r, term, info = env.step({i: int(d["actions"][t, i]) for i in range(4)})
        for i in range(4):
            assert (o[i].cpu().numpy() == d["obs"][t, i]).all(), (t, i)
        assert r[0] == d["reward"][t, 0] and r[1] == d["reward"][t, 1] and r[2] == r[0] and r[3] == r[1]
        assert all(v == bool(d["done"][t]) for v in term.values())
        assert info["score_change"] == d["score_change"][t]
        if t < len(meta["legal_lists_first_ticks"]):
            assert [info["legal_actions"][i] for i in range(4)] == meta["legal_lists_first_ticks"][t]
        assert env.game.state.data.score == d["sub_score"][t, 3]
        if d["resets"][t]:
            env.reset()
    env.close()


@pytest.mark.parametrize("fixture", ["bots_tiny_baselineTeam.json", "bots_small_baselineTeam.json",
                                     "bots_small_randomTeam.json", "bots_tiny_randomTeam.json"])
def test_bots_stream_exact(tmp_path, fixture):
    import pmx
    from pmx.game_state import DIR_CODE
    g = G.load_json(fixture)
    path = _write_layout(tmp_path, g["layout"], "l.lay")
    random.seed(g["random_seed"])
    env = pmx.gymPacMan_parallel_env(layout_file=path, length=g["length"], self_play=False, enemieName=g["team"])
    assert not isinstance(env.agents[0], int) and not isinstance(env.agents[2], int) and env.agents[1] == 1
    env.reset()
    chosen = []

    def hook():
        for b in (env.agents[0], env.agents[2]):
            orig = b.getAction

            def wrap(gs, _o=orig):
                a = _o(gs)
                chosen.append(a)
                return a
            b.getAction = wrap
    hook()
    for t, (a1, a3) in enumerate(g["blue_actions"]):
        chosen.clear()
        o, r, term, info = env.step({env.agents[1]: a1, env.agents[3]: a3})
        assert [DIR_CODE[c] for c in chosen] == g["red_actions"][t], f"tick {t}: bot actions diverged"
        assert env.game.state.data.score == g["scores"][t], t
        assert [r[env.agents[0]], r[env.agents[1]]] == g["rewards"][t], t
        done = any(term.values())
        assert done == g["dones"][t], t
        if done:
            env.reset()
            hook()
    env.close()
