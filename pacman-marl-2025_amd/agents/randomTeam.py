"""randomTeam: both agents pick uniformly among their legal actions (agents/randomTeam.py:90-100 of the reference)."""
import random

from pmx.capture_agents import CaptureAgent


def createTeam(firstIndex, secondIndex, isRed, first="DummyAgent", second="DummyAgent"):
    return [DummyAgent(firstIndex), DummyAgent(secondIndex)]


class DummyAgent(CaptureAgent):
    def chooseAction(self, gameState):
        return random.choice(gameState.getLegalActions(self.index))
