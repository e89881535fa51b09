"""CPU suite: the trainer's pure tensor helpers against the golden fixtures (no GPU needed)."""
import numpy as np
import torch

import _golden as G


def _words(obs_agent):
    """x | y << 8 | carry << 16 from an observation's plane 1 (what pmx_step_out.agent_dev carries)."""
    p1 = obs_agent[1]
    y, x = np.argwhere(p1 > 0)[0]
    return int(x) | (int(y) << 8) | ((int(p1[y, x]) - 1) << 16)


def test_shaping_from_agent_words_matches_reference_shaping():
    from pmx import trainer
    d, meta = G.load("shaping.npz")
    tr, _ = G.load(meta["traj"])
    obs = tr["obs"]
    T = int(d["T"])
    w = np.array([[_words(obs[t, i]) for i in range(4)] for t in range(T)], np.int32)
    got = trainer.shaping_from_agent_words(torch.tensor(w[:-1]), torch.tensor(w[1:])).numpy()
    assert got.tobytes() == d["shaping"][:T - 1].tobytes()      # float64, bit-equal to compute_heuristic_shaping


def test_canonicalize_and_merge_match_reference():
    from pmx import trainer
    d, meta = G.load("shaping.npz")
    tr, _ = G.load(meta["traj"])
    for j, t in enumerate(d["ticks"]):
        o = torch.tensor(tr["obs"][t]).float()
        canon = trainer.canonicalize_obs(o)
        assert (canon.numpy() == d["canon_red"][j]).all()
        assert (trainer.merge_obs(o[1:2], o[3:4])[0].numpy() == d["merged_blue"][j]).all()
        assert (trainer.merge_obs(canon[0:1], canon[2:3])[0].numpy() == d["merged_red"][j]).all()
