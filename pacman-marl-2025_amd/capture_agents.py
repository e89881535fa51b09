"""CaptureAgent action API (captureAgents.py:47-304) for host-side bots on top of the GPU env.

  createTeam(firstIndex, secondIndex, isRed, **kw) -> [agent, agent]      module-level factory of a team file
  agent.registerInitialState(gameState)                                   captureAgents.py:91-109
  agent.getAction(gameState) -> 'North' | 'South' | 'East' | 'West' | 'Stop'   captureAgents.py:144-162
The Distancer is fed by pmx_maze_distances (distanceCalculator.py:111-150 on the GPU), computed once per layout
instead of once per bot per reset.
"""
import importlib.util
import os

from .game_state import Directions, GameState  # noqa: F401  (team files import these from here)

_DIST_CACHE = {}


def nearestPoint(pos):                                              # util.py:513-521
    return (int(pos[0] + 0.5), int(pos[1] + 0.5))


def manhattanDistance(a, b):                                        # util.py:205-207
    return abs(a[0] - b[0]) + abs(a[1] - b[1])


class Distancer:
    """distanceCalculator.Distancer (distanceCalculator.py:24-75): all-pairs maze distances of a layout."""

    def __init__(self, layout, maze_distance_fn, default=10000):
        self.layout, self.default, self._fn = layout, default, maze_distance_fn
        self._distances = None

    def getMazeDistances(self):
        key = tuple(self.layout.text)
        if key not in _DIST_CACHE:
            cells, dist = self._fn()
            cells = [tuple(c) for c in cells.cpu().tolist()]
            d = dist.cpu().numpy()
            _DIST_CACHE[key] = ({c: i for i, c in enumerate(cells)}, d)
        self._distances = _DIST_CACHE[key]

    def getDistance(self, pos1, pos2):
        if self._distances is None:
            return manhattanDistance(pos1, pos2)
        index, d = self._distances
        p1 = (int(pos1[0]), int(pos1[1]))
        p2 = (int(pos2[0]), int(pos2[1]))
        if p1 not in index or p2 not in index:
            raise Exception("Positions not in grid: " + str((pos1, pos2)))   # distanceCalculator.py:57-62
        v = int(d[index[p1], index[p2]])
        return v if v != 255 else 2 ** 63 - 1                       # sys.maxsize for unreachable pairs


class CaptureAgent:
    def __init__(self, index, timeForComputing=.1):
        self.index = index
        self.red = None
        self.agentsOnTeam = None
        self.distancer = None
        self.observationHistory = []
        self.timeForComputing = timeForComputing
        self.display = None

    def registerInitialState(self, gameState):
        self.red = gameState.isOnRedTeam(self.index)
        self.distancer = Distancer(gameState._layout, gameState._engine.env.maze_distances)
        self.distancer.getMazeDistances()

    def final(self, gameState):
        self.observationHistory = []

    def registerTeam(self, agentsOnTeam):
        self.agentsOnTeam = agentsOnTeam

    def getAction(self, gameState):
        self.observationHistory.append(gameState)
        myPos = gameState.getAgentState(self.index).getPosition()
        if myPos != nearestPoint(myPos):
            return gameState.getLegalActions(self.index)[0]
        return self.chooseAction(gameState)

    def chooseAction(self, gameState):
        raise NotImplementedError

    # convenience getters, captureAgents.py:175-248
    def getFood(self, gameState):
        return gameState.getBlueFood() if self.red else gameState.getRedFood()

    def getFoodYouAreDefending(self, gameState):
        return gameState.getRedFood() if self.red else gameState.getBlueFood()

    def getCapsules(self, gameState):
        return gameState.getBlueCapsules() if self.red else gameState.getRedCapsules()

    def getCapsulesYouAreDefending(self, gameState):
        return gameState.getRedCapsules() if self.red else gameState.getBlueCapsules()

    def getOpponents(self, gameState):
        return gameState.getBlueTeamIndices() if self.red else gameState.getRedTeamIndices()

    def getTeam(self, gameState):
        return gameState.getRedTeamIndices() if self.red else gameState.getBlueTeamIndices()

    def getScore(self, gameState):
        return gameState.getScore() if self.red else gameState.getScore() * -1

    def getMazeDistance(self, pos1, pos2):
        return self.distancer.getDistance(pos1, pos2)

    def getPreviousObservation(self):
        return None if len(self.observationHistory) == 1 else self.observationHistory[-2]

    def getCurrentObservation(self):
        return self.observationHistory[-1]


_AGENT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "agents")


def load_agents(is_red, factory):
    """capture.loadAgents (capture.py:914-950): import a team file and call createTeam(i0, i1, isRed).
    `factory` is a path or a bare team name looked up in this package's agents/ directory."""
    path = factory if factory.endswith(".py") else factory + ".py"
    if not os.path.exists(path):
        path = os.path.join(_AGENT_DIR, os.path.basename(path))
    spec = importlib.util.spec_from_file_location("player" + str(int(is_red)), path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    add = 0 if is_red else 1
    return module.createTeam(0 + add, 2 + add, is_red)
