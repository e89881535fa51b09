"""baselineTeam for the GPU drop-in env: one food-seeking and one home-guarding reflex bot.

Behaviour to match (pinned stream-exactly by fixture G9, tests/test_gpu_dropin.py::test_bots_stream_exact): the reference
team (agents/baselineTeam.py:34-187) scores the successor of every legal action with an integer-weighted feature sum and
plays `random.choice` of the best ones -- one draw per turn from the GLOBAL `random` stream -- except when the food it
hunts is gone, in which case it walks to the legal successor nearest its start cell (first minimum, no draw).

This file is written against the pmx API, not the reference's class tree: a bot asks the env for the five raw successor
records of its agent in ONE GPU round trip (SuccessorEngine.successors -> pmx_successor) and scores them as integers
straight from the packed state (bit rows, position bytes), the same arithmetic `bot_action<I>` in csrc/pmx_step.hip
uses in-kernel.  Only what the outside world sees is kept: `createTeam(...)`'s signature and role names, the
CaptureAgent entry points, the action strings, and the order of `random` calls.
"""
import random

from pmx.capture_agents import CaptureAgent
from pmx.game_state import DIR_CODE

_BACKWARDS = (2, 3, 0, 1, 4)          # action code that undoes a heading (North<->South, East<->West, Stop)
_STOP = 4
_FAR = 1 << 60


def _cells(rows, height, mask):
    """(x, y) of every set bit of `rows[y] & mask`."""
    for y in range(height):
        r = int(rows[y]) & mask
        while r:
            low = r & -r
            yield low.bit_length() - 1, y
            r ^= low


def _forage_score(bot, now, nxt, code):
    """100 per pellet fewer on the far side, minus the maze distance to the nearest one left."""
    me = (int(nxt.pos[bot.index][0]), int(nxt.pos[bot.index][1]))
    left, nearest = 0, _FAR
    for cell in _cells(nxt.food, bot.height, bot.prey_mask):
        left += 1
        d = bot.getMazeDistance(me, cell)
        if d < nearest:
            nearest = d
    return -100 * left - (nearest if left else 0)


def _guard_score(bot, now, nxt, code):
    """Stay a ghost (+100), -1000 per invader, close in on the nearest (-10 per step), dislike stopping and turning back."""
    me = (int(nxt.pos[bot.index][0]), int(nxt.pos[bot.index][1]))
    total = 0 if nxt.pac[bot.index] else 100
    gap = _FAR
    for foe in bot.foes:
        if nxt.pac[foe]:
            total -= 1000
            gap = min(gap, bot.getMazeDistance(me, (int(nxt.pos[foe][0]), int(nxt.pos[foe][1]))))
    if gap != _FAR:
        total -= 10 * gap
    if code == _STOP:
        total -= 100
    if code == _BACKWARDS[now.dir[bot.index]]:
        total -= 2
    return total


_ROLES = {"OffensiveReflexAgent": _forage_score, "DefensiveReflexAgent": _guard_score}


class ReflexBot(CaptureAgent):
    def __init__(self, index, score):
        CaptureAgent.__init__(self, index)
        self.score = score

    def registerInitialState(self, gameState):
        CaptureAgent.registerInitialState(self, gameState)
        lay = gameState._layout
        self.home = gameState.getInitialAgentPosition(self.index)
        self.height = lay.height
        left_half = (1 << (lay.width // 2)) - 1
        self.prey_mask = (((1 << lay.width) - 1) & ~left_half) if self.red else left_half
        self.foes = self.getOpponents(gameState)

    def chooseAction(self, gameState):
        now = gameState._state
        names = gameState.getLegalActions(self.index)                      # reference list order: N, S, E, W, Stop
        options = gameState._engine.successors(now, self.index)            # all five successors, one GPU round trip
        if not any(int(now.food[y]) & self.prey_mask for y in range(self.height)):
            best_name, best_d = None, 9999
            for name in names:
                nxt = options[DIR_CODE[name]][0]
                d = self.getMazeDistance(self.home, (int(nxt.pos[self.index][0]), int(nxt.pos[self.index][1])))
                if d < best_d:
                    best_name, best_d = name, d
            return best_name
        marks = [self.score(self, now, options[DIR_CODE[name]][0], DIR_CODE[name]) for name in names]
        top = max(marks)
        return random.choice([name for name, m in zip(names, marks) if m == top])


def createTeam(firstIndex, secondIndex, isRed, first="OffensiveReflexAgent", second="DefensiveReflexAgent"):
    return [ReflexBot(firstIndex, _ROLES[first]), ReflexBot(secondIndex, _ROLES[second])]
