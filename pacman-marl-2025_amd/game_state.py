"""Host-side view of one game for CaptureAgent bots: the accessor surface of capture.GameState (capture.py:101-234)
over a pmx_state copied from the GPU.

Every rule evaluation stays on the GPU: legal-action masks come from pmx_observe, successors from pmx_successor on
a small scratch handle (all five actions of an agent are generated in ONE batch and cached, because reflex bots ask
for every legal action's successor: agents/baselineTeam.py:65-104).  What runs on the host is only data formatting:
unpacking bit rows into Grid-like objects, splitting them at the half line (capture.py:332-350).
"""
import ctypes as C

import torch

from . import _lib
from .vec_env import LEGAL_LIST_ORDER, PmxVecEnv

DIR_NAMES = ("North", "East", "South", "West", "Stop")            # action / direction codes 0..4
DIR_CODE = {n: i for i, n in enumerate(DIR_NAMES)}


class Directions:                                                  # game.py:49-68
    NORTH, SOUTH, EAST, WEST, STOP = "North", "South", "East", "West", "Stop"
    LEFT = {NORTH: WEST, SOUTH: EAST, EAST: NORTH, WEST: SOUTH, STOP: STOP}
    RIGHT = {v: k for k, v in LEFT.items()}
    REVERSE = {NORTH: SOUTH, SOUTH: NORTH, EAST: WEST, WEST: EAST, STOP: STOP}


class Grid:
    """Read-mostly stand-in for game.Grid (game.py:162-278): g[x][y], asList, count, width, height, data."""

    def __init__(self, width, height, rows=None):
        self.width, self.height = width, height
        self.data = [[False] * height for _ in range(width)]
        if rows is not None:
            for y in range(height):
                r = int(rows[y])
                x = 0
                while r:
                    if r & 1:
                        self.data[x][y] = True
                    r >>= 1
                    x += 1

    def __getitem__(self, i):
        return self.data[i]

    def __eq__(self, other):
        return other is not None and self.data == other.data

    def asList(self, key=True):                                    # x outer, y inner (game.py:225-230)
        return [(x, y) for x in range(self.width) for y in range(self.height) if self.data[x][y] == key]

    def count(self, item=True):
        return sum(col.count(item) for col in self.data)

    def copy(self):
        g = Grid(self.width, self.height)
        g.data = [col[:] for col in self.data]
        return g


class Configuration:                                               # game.py:70-118
    def __init__(self, pos, direction):
        self.pos, self.direction = pos, direction

    def getPosition(self):
        return self.pos

    def getDirection(self):
        return self.direction


class AgentState:                                                  # game.py:120-160
    def __init__(self, pos, direction, is_pacman, scared, carrying, returned, start):
        self.configuration = Configuration(pos, direction)
        self.start = Configuration(start, Directions.STOP)
        self.isPacman, self.scaredTimer = is_pacman, scared
        self.numCarrying, self.numReturned = carrying, returned

    def getPosition(self):
        return self.configuration.getPosition()

    def getDirection(self):
        return self.configuration.getDirection()


class _LayoutView:
    """data.layout.{width,height,walls,agentPositions} as the bots read them."""

    def __init__(self, layout):
        self.width, self.height = layout.width, layout.height
        self.walls = Grid(layout.width, layout.height, layout.wall_rows)
        self.agentPositions = [(False, p) for p in layout.agent_positions]
        self.totalFood = layout.total_food

    def isWall(self, pos):
        return self.walls[pos[0]][pos[1]]


class _Data:
    pass


class SuccessorEngine:
    """Scratch handle (8 envs) that answers generateSuccessor for all five actions of one agent in one GPU round trip."""

    def __init__(self, layout, device):
        self.env = PmxVecEnv(layout, 8, length=1 << 30, auto_reset=False, device=device)
        self.actions = torch.tensor([0, 1, 2, 3, 4, 4, 4, 4], dtype=torch.int8, device=self.env.device)

    def successors(self, state, agent):
        arr = (_lib.State * 8)(*([state] * 8))
        self.env.set_state(arr)
        sc = self.env.successor(agent, self.actions)
        _, legal = self.env.observe(want_obs=False)
        out = self.env.get_state(0, 5)
        sc = sc.cpu().tolist()
        legal = legal.cpu().tolist()
        return [(_copy_state(out[a]), legal[a], sc[a]) for a in range(5)]


def _copy_state(s):
    c = _lib.State()
    C.memmove(C.byref(c), C.byref(s), C.sizeof(_lib.State))
    return c


class GameState:
    """capture.GameState accessors (capture.py:101-234) over (pmx_state, legal masks)."""

    def __init__(self, layout, layout_view, state, legal, engine, timeleft=0, score_change=0):
        self._layout, self._state, self._legal, self._engine = layout, state, legal, engine
        self._succ = {}
        W, H = layout.width, layout.height
        self.data = _Data()
        self.data.layout = layout_view
        self.data.score = int(state.score)
        self.data.scoreChange = score_change
        self.data.timeleft = timeleft
        self.data.food = Grid(W, H, state.food)
        self.data.capsules = [(x, y) for y in range(H) for x in range(W) if (int(state.caps[y]) >> x) & 1]
        self.data._win = False
        self.data.agentStates = [
            AgentState((int(state.pos[i][0]), int(state.pos[i][1])), DIR_NAMES[state.dir[i]], bool(state.pac[i]),
                       int(state.scared[i]), int(state.carry[i]), int(state.ret[i]), layout.agent_positions[i])
            for i in range(4)]
        self.redTeam, self.blueTeam = [0, 2], [1, 3]                # capture.py:316-319 (validated by pmx_create)
        self.teams = [True, False, True, False]
        self.agentDistances = []

    # -- accessors ------------------------------------------------------------------------------------------------
    def getLegalActions(self, agentIndex=0):                        # capture.py:101-105, list order N,S,E,W,Stop
        m = self._legal[agentIndex]
        return [DIR_NAMES[a] for a in LEGAL_LIST_ORDER if (m >> a) & 1]

    def generateSuccessor(self, agentIndex, action):                # capture.py:107-123
        if agentIndex not in self._succ:
            self._succ[agentIndex] = self._engine.successors(self._state, agentIndex)
        st, legal, sc = self._succ[agentIndex][DIR_CODE[action] if isinstance(action, str) else int(action)]
        return GameState(self._layout, self.data.layout, st, legal, self._engine, self.data.timeleft - 1, sc)

    def getAgentState(self, index):
        return self.data.agentStates[index]

    def getAgentPosition(self, index):
        return self.data.agentStates[index].getPosition()

    def getNumAgents(self):
        return 4

    def getScore(self):
        return self.data.score

    def _half(self, red):                                           # halfGrid, capture.py:332-342
        W, H, half = self._layout.width, self._layout.height, self._layout.width // 2
        lo = (1 << half) - 1
        mask = lo if red else (((1 << W) - 1) & ~lo)
        return Grid(W, H, [int(self._state.food[y]) & mask for y in range(H)])

    def getRedFood(self):
        return self._half(True)

    def getBlueFood(self):
        return self._half(False)

    def getRedCapsules(self):                                       # halfList, capture.py:344-350
        half = self._layout.width / 2
        return [(x, y) for (x, y) in self.data.capsules if x <= half]

    def getBlueCapsules(self):
        half = self._layout.width / 2
        return [(x, y) for (x, y) in self.data.capsules if x > half]

    def getCapsules(self):
        return self.data.capsules

    def getWalls(self):
        return self.data.layout.walls

    def hasFood(self, x, y):
        return self.data.food[x][y]

    def hasWall(self, x, y):
        return self.data.layout.walls[x][y]

    def isOver(self):
        return self.data._win

    def getRedTeamIndices(self):
        return self.redTeam[:]

    def getBlueTeamIndices(self):
        return self.blueTeam[:]

    def isOnRedTeam(self, agentIndex):
        return self.teams[agentIndex]

    def getAgentDistances(self):
        return None

    def getInitialAgentPosition(self, agentIndex):
        return self._layout.agent_positions[agentIndex]

    def isRed(self, configOrPos):                                   # capture.py:325-330
        pos = configOrPos if isinstance(configOrPos, tuple) else configOrPos.pos
        return pos[0] < self._layout.width / 2
