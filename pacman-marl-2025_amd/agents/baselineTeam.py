"""baselineTeam: one offensive and one defensive reflex agent (agents/baselineTeam.py:34-187 of the reference).

Both score every legal action by a linear function of features of the successor state and pick uniformly among the
best; the draw uses the stdlib `random` module in the reference's call order, so with the same seed the action stream
is the reference's (fixture G9)."""
import random

from pmx.capture_agents import CaptureAgent, Counter, Directions, nearestPoint


def createTeam(firstIndex, secondIndex, isRed, first="OffensiveReflexAgent", second="DefensiveReflexAgent"):
    kinds = {"OffensiveReflexAgent": OffensiveReflexAgent, "DefensiveReflexAgent": DefensiveReflexAgent}
    return [kinds[first](firstIndex), kinds[second](secondIndex)]


class ReflexCaptureAgent(CaptureAgent):
    def registerInitialState(self, gameState):
        self.start = gameState.getAgentPosition(self.index)
        CaptureAgent.registerInitialState(self, gameState)

    def chooseAction(self, gameState):                               # :65-92
        actions = gameState.getLegalActions(self.index)
        values = [self.evaluate(gameState, a) for a in actions]
        best = max(values)
        bestActions = [a for a, v in zip(actions, values) if v == best]
        if len(self.getFood(gameState).asList()) <= 0:               # nothing left to eat: head home
            bestDist, bestAction = 9999, None
            for action in actions:
                pos2 = self.getSuccessor(gameState, action).getAgentPosition(self.index)
                dist = self.getMazeDistance(self.start, pos2)
                if dist < bestDist:
                    bestAction, bestDist = action, dist
            return bestAction
        return random.choice(bestActions)

    def getSuccessor(self, gameState, action):                       # :94-104
        successor = gameState.generateSuccessor(self.index, action)
        pos = successor.getAgentState(self.index).getPosition()
        if pos != nearestPoint(pos):
            return successor.generateSuccessor(self.index, action)
        return successor

    def evaluate(self, gameState, action):
        return self.getFeatures(gameState, action) * self.getWeights(gameState, action)

    def getFeatures(self, gameState, action):
        f = Counter()
        f["successorScore"] = self.getScore(self.getSuccessor(gameState, action))
        return f

    def getWeights(self, gameState, action):
        return {"successorScore": 1.0}


class OffensiveReflexAgent(ReflexCaptureAgent):                      # :125-151
    def getFeatures(self, gameState, action):
        f = Counter()
        successor = self.getSuccessor(gameState, action)
        foodList = self.getFood(successor).asList()
        f["successorScore"] = -len(foodList)
        if len(foodList) > 0:
            myPos = successor.getAgentState(self.index).getPosition()
            f["distanceToFood"] = min(self.getMazeDistance(myPos, food) for food in foodList)
        return f

    def getWeights(self, gameState, action):
        return {"successorScore": 100, "distanceToFood": -1}


class DefensiveReflexAgent(ReflexCaptureAgent):                      # :153-187
    def getFeatures(self, gameState, action):
        f = Counter()
        successor = self.getSuccessor(gameState, action)
        myState = successor.getAgentState(self.index)
        myPos = myState.getPosition()
        f["onDefense"] = 0 if myState.isPacman else 1
        enemies = [successor.getAgentState(i) for i in self.getOpponents(successor)]
        invaders = [a for a in enemies if a.isPacman and a.getPosition() is not None]
        f["numInvaders"] = len(invaders)
        if invaders:
            f["invaderDistance"] = min(self.getMazeDistance(myPos, a.getPosition()) for a in invaders)
        if action == Directions.STOP:
            f["stop"] = 1
        if action == Directions.REVERSE[gameState.getAgentState(self.index).configuration.direction]:
            f["reverse"] = 1
        return f

    def getWeights(self, gameState, action):
        return {"numInvaders": -1000, "onDefense": 100, "invaderDistance": -10, "stop": -100, "reverse": -2}
