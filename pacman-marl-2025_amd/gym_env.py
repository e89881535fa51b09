"""Drop-in for the reference's gymPacMan.gymPacMan_parallel_env (gymPacMan.py:12-270) on one GPU-resident game.

Same constructor arguments, same `agents` keys (the int i for learner slots, the bot OBJECT for red slots when
self_play=False), same reset()/step() return structure (dicts keyed by env.agents[i], float32 [8,H,W] tensors, float
rewards, legal-action lists in the reference's order), same quirks (staggered observations, length+1 ticks, illegal
action -> Stop, the startingIndex draw from the global `random` at every newGame).  The tick itself runs in
libpmx_hip.so; red bots act on the mid-tick state through pmx_step_agent (gymPacMan.py:157).
"""
import os
import random

import torch

from . import maze_generator
from .capture_agents import load_agents
from .game_state import DIR_CODE, DIR_NAMES, GameState, SuccessorEngine, _LayoutView
from .layout import Layout, get_layout
from .vec_env import PmxVecEnv, legal_list

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Game:
    """env.game: callers read env.game.state (pacman_mappo_resnet.py:529: env.game.state.data.score)."""

    def __init__(self, env):
        self._env = env
        self.length = env.length

    @property
    def state(self):
        return self._env._game_state()


class gymPacMan_parallel_env:
    metadata = {"render_modes": ["human"], "name": "rps_v2"}

    def __init__(self, layout_file=os.path.join(_HERE, "layouts", "smallCapture.lay"), display=False, length=299,
                 reward_forLegalAction=True, defenceReward=True, random_layout=False, enemieName="randomTeam",
                 self_play=False, device="cuda:0"):
        if display:
            raise NotImplementedError("graphics display is out of scope of the MI355X env (DESIGN.md section 7)")
        self.device = device
        if os.path.exists(layout_file) and not random_layout:           # gymPacMan.py:16-21
            self.layout = Layout.from_file(layout_file)
        elif get_layout(layout_file) is not None and not random_layout:
            self.layout = get_layout(layout_file)
        else:
            self.layout = Layout.from_text(maze_generator.random_layout())
        self.random_layout = random_layout
        self.enemieName = enemieName
        self.self_play = self_play
        self.display = None
        self.reward_forLegalAction = reward_forLegalAction
        self.defenceReward = defenceReward
        self.length = length
        self.num_moves = 0
        self.steps = 0
        self.action_mapping = {0: "North", 1: "East", 2: "South", 3: "West", 4: "Stop", "North": "North",
                               "East": "East", "South": "South", "West": "West", "Stop": "Stop"}
        self.reversed_action_mapping = {"North": 0, "East": 1, "South": 2, "West": 3, "Stop": 4, 0: 0, 1: 1, 2: 2, 3: 3, 4: 4}
        self._bots = load_agents(True, self.enemieName)
        self._env = None
        self._new_game(first_bots=[0, 1])
        self.max_score = self._game_state().getBlueFood().count()       # gymPacMan.py:64

    # -- construction of a game (CaptureRules.newGame, capture.py:369-382) ---------------------------------------------
    def _new_game(self, first_bots):
        new_agents = [None] * 4
        if not self.self_play:
            for i in first_bots:
                new_agents[self._bots[i].index] = self._bots[i]
        self.agents = [a if a is not None else i for i, a in enumerate(new_agents)]
        if self._env is None or self._env.layout is not self.layout:
            if self._env is not None:
                self._env.close()
                self._engine.env.close()
            self._env = PmxVecEnv(self.layout, 1, length=self.length, reward_forLegalAction=self.reward_forLegalAction,
                                  defenceReward=self.defenceReward, auto_reset=False, obs_dtype="float32", device=self.device)
            self._engine = SuccessorEngine(self.layout, self.device)
            self._layout_view = _LayoutView(self.layout)
        random.randint(0, 1)                                            # `starter`, drawn and unused (capture.py:372)
        self._env.reset(want_obs=False)
        self._cached_state = None
        self._sub = 0
        self.game = _Game(self)
        for agent in self.agents:
            if not isinstance(agent, int):
                agent.registerInitialState(self._game_state())

    def _game_state(self):
        if self._cached_state is None:
            st = self._env.get_state(0, 1)[0]
            _, legal = self._env.observe(want_obs=False)
            timeleft = self.length - 4 * int(st.steps) - self._sub      # one per generateSuccessor (capture.py:122)
            self._cached_state = GameState(self.layout, self._layout_view, st, legal[0].cpu().tolist(), self._engine, timeleft)
        return self._cached_state

    # -- gymPacMan surface ------------------------------------------------------------------------------------------------
    def reset(self, layout_file="None", enemieName="None"):             # gymPacMan.py:92-141
        self.steps = 0
        if os.path.exists(layout_file):
            self.layout = Layout.from_file(layout_file)
        if self.random_layout:
            self.layout = Layout.from_text(maze_generator.random_layout())
        first = [0, 2]
        if enemieName != "None":
            self.enemieName = enemieName
            self._bots = load_agents(True, self.enemieName)
            first = [0, 1]
        if not self.self_play and enemieName == "None":
            # the reference indexes its previous `agents` list with [0, 2]: the two bot objects sit in slots 0 and 2
            self._bots = [self.agents[0], self.agents[2]]
            first = [0, 1]
        self.num_moves = 0
        self._new_game(first_bots=first)
        obs, legal = self._env.reset(want_obs=True)
        self._cached_state = None
        observations = {self.agents[i]: obs[0, i].clone() for i in range(4)}
        lg = legal[0].cpu().tolist()
        return observations, {"legal_actions": {self.agents[i]: legal_list(lg[i]) for i in range(4)}}

    def step(self, actions):                                            # gymPacMan.py:143-193
        env = self._env
        observations = {}
        for i in range(4):
            if i in (1, 3) or self.self_play:
                a = self.action_mapping[actions[self.agents[i]]]        # KeyError on an unknown action, as in the reference
            else:
                self._cached_state = None
                a = self.agents[i].getAction(self._game_state())        # the bot sees the full mid-tick state (:157)
            code = torch.tensor([DIR_CODE[a]], dtype=torch.int8, device=env.device)
            observations[self.agents[i]] = env.step_agent(i, code).clone()[0]
            self._sub = (i + 1) & 3
        self._cached_state = None
        rew = env.reward[0].cpu().tolist()
        done = bool(env.done[0].item())
        rewards = {self.agents[i]: (rew[0] if i in (0, 2) else rew[1]) for i in range(4)}
        terminations = {agent: done for agent in self.agents}
        self.steps += 1
        lg = env.legal[0].cpu().tolist()
        return observations, rewards, terminations, {
            "legal_actions": {self.agents[i]: legal_list(lg[i]) for i in range(4)},
            "score_change": int(env.score_change[0].item())}

    def get_Observation(self, agentIndex):                              # gymPacMan.py:195-229
        obs, _ = self._env.observe(want_obs=True, want_legal=False)
        return obs[0, agentIndex].clone()

    def check_termination(self):                                        # gymPacMan.py:261-270 on the current state
        gs = self._game_state()
        done = False
        if gs.getBlueFood().count() == 0 and gs.getAgentState(0).numCarrying == 0 and gs.getAgentState(2).numCarrying == 0:
            done = True
        if gs.getRedFood().count() == 0 and gs.getAgentState(1).numCarrying == 0 and gs.getAgentState(3).numCarrying == 0:
            done = True
        if self.steps >= self.length:
            done = True
        return {agent: done for agent in self.agents}

    def close(self):
        if self._env is not None:
            self._env.close()
            self._engine.env.close()
            self._env = None
