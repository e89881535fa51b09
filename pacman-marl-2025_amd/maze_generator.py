"""Random capture mazes on the host: mazeGenerator.py restated with the same stdlib `random` call order.

The reference draws every random number from the GLOBAL `random` module after `random.seed(seed)`
(mazeGenerator.py:255-264), so a maze is a pure function of the seed as long as the sequence of
randint / random / gauss / choice / shuffle calls is the same.  This module reproduces that sequence on a private
`random.Random(seed)` (CPython's Mersenne Twister and its derived methods are the same code), which leaves the
caller's global stream alone; `generate_maze(seed, use_global=True)` reseeds the global module instead, for callers
that rely on the reference's side effect on later bot choices ("RNG").

Pinned by tests/golden/mazes.json (G5).
"""
import random as _random

W, F, C, E = "%", ".", "o", " "


class _Maze:
    """mazeGenerator.py:42-118"""

    def __init__(self, rows, cols, anchor=(0, 0), root=None, rng=None):
        self.r, self.c = rows, cols
        self.grid = [[E for _ in range(cols)] for _ in range(rows)]
        self.anchor = anchor
        self.rooms = []
        self.root = root if root is not None else self
        self.rng = rng if rng is not None else self.root.rng

    def to_map(self):   # :57-75 mirrored copy on the right, then a border
        for row in range(self.r):
            for col in range(self.c - 1, -1, -1):
                self.grid[row].append(self.grid[row][col])
        self.c *= 2
        for row in range(self.r):
            self.grid[row] = [W] + self.grid[row] + [W]
        self.c += 2
        self.grid.insert(0, [W for _ in range(self.c)])
        self.grid.append([W for _ in range(self.c)])
        self.r += 2

    def __str__(self):
        return "\n".join("".join(self.grid[row][col] for col in range(self.c)) for row in range(self.r))

    def add_wall(self, i, gaps=1, vert=True):   # :85-118
        add_r, add_c = self.anchor
        root = self.root
        if vert:
            gaps = min(self.r, gaps)
            slots = [add_r + x for x in range(self.r)]
            if 0 not in slots:
                if root.grid[min(slots) - 1][add_c + i] == E:
                    slots.remove(min(slots))
                if len(slots) <= gaps:
                    return 0
            if root.c - 1 not in slots and max(slots) + 1 < len(root.grid):
                if root.grid[max(slots) + 1][add_c + i] == E:
                    slots.remove(max(slots))
            if len(slots) <= gaps:
                return 0
            self.rng.shuffle(slots)
            for row in slots[int(round(gaps)):]:
                root.grid[row][add_c + i] = W
            self.rooms.append(_Maze(self.r, i, (add_r, add_c), root))
            self.rooms.append(_Maze(self.r, self.c - i - 1, (add_r, add_c + i + 1), root))
        else:
            gaps = min(self.c, gaps)
            slots = [add_c + x for x in range(self.c)]
            if 0 not in slots:
                if root.grid[add_r + i][min(slots) - 1] == E:
                    slots.remove(min(slots))
                if len(slots) <= gaps:
                    return 0
            if root.r - 1 not in slots and max(slots) + 1 < len(root.grid[0]):
                if root.grid[add_r + i][max(slots) + 1] == E:
                    slots.remove(max(slots))
            if len(slots) <= gaps:
                return 0
            self.rng.shuffle(slots)
            for col in slots[int(round(gaps)):]:
                root.grid[add_r + i][col] = W
            self.rooms.append(_Maze(i, self.c, (add_r, add_c), root))
            self.rooms.append(_Maze(self.r - i - 1, self.c, (add_r + i + 1, add_c), root))
        return 1


def _make(room, depth, gaps=1, vert=True, min_width=1, gapfactor=0.5):   # :153-184
    if room.r <= min_width and room.c <= min_width:
        return
    num = room.c if vert else room.r
    if num < min_width + 2:
        vert = not vert
        num = room.c if vert else room.r
    wall_slots = [num - 2] if depth == 0 else range(1, num - 1)
    if len(wall_slots) == 0:
        return
    choice = room.rng.choice(wall_slots)
    if not room.add_wall(choice, gaps, vert):
        return
    for sub in room.rooms:
        _make(sub, depth + 1, max(1, gaps * gapfactor), not vert, min_width, gapfactor)


def _make_with_prison(room, depth, gaps=1, vert=True, min_width=1, gapfactor=0.5):   # :120-151
    rng = room.rng
    rng.randint(0, 2)       # drawn and overwritten in the reference (:124)
    rng.random()            # proll (:125); the result is discarded because p is forced to 0 (:134)
    p = 0
    add_r, add_c = room.anchor
    room.rooms.append(_Maze(room.r, room.c - (2 * p), (add_r, add_c + (2 * p)), room.root))
    for sub in room.rooms:
        _make(sub, depth + 1, gaps, vert, min_width, gapfactor)
    return 2 * p


def _add_pacman_stuff(maze, max_food=60, max_capsules=4, toskip=0):   # :194-251
    rng = maze.rng
    max_depth = 2
    depth = 0
    total_food = 0
    while True:
        new_grid = [row[:] for row in maze.grid]
        depth += 1
        num_added = 0
        for row in range(1, maze.r - 1):
            for col in range(1 + toskip, (maze.c // 2) - 1):
                if (row > maze.r - 6) and (col < 6):
                    continue
                if maze.grid[row][col] != E:
                    continue
                neighbors = ((maze.grid[row - 1][col] == E) + (maze.grid[row][col - 1] == E)
                             + (maze.grid[row + 1][col] == E) + (maze.grid[row][col + 1] == E))
                if neighbors == 1:
                    new_grid[row][col] = F
                    new_grid[row][maze.c - col - 1] = F
                    num_added += 2
                    total_food += 2
        maze.grid = new_grid
        if num_added == 0:
            break
        if depth >= max_depth:
            break
    maze.grid[1][1] = "3"
    maze.grid[2][1] = "1"
    maze.grid[1][maze.c - 2] = "4"
    maze.grid[2][maze.c - 2] = "2"
    total_capsules = 0
    draws = 0
    while total_capsules < max_capsules:
        draws += 1
        if draws > 200000:   # the reference loops forever when no cell qualifies (only possible for non-reference sizes)
            raise ValueError("maze size leaves no legal capsule cell")
        row = rng.randint(1, maze.r - 1)
        col = rng.randint(1 + toskip, (maze.c // 2) - 2)
        if (row > maze.r - 6) and (col < 6):
            continue
        if abs(col - maze.c / 2) < 3:
            continue
        if maze.grid[row][col] == E:
            maze.grid[row][col] = C
            maze.grid[row][maze.c - col - 1] = C
            total_capsules += 2
    draws = 0
    while total_food < max_food:
        draws += 1
        if draws > 2000000:
            raise ValueError("maze size leaves too few legal food cells")
        row = rng.randint(1, maze.r - 1)
        col = rng.randint(1 + toskip, (maze.c // 2) - 1)
        if (row > maze.r - 6) and (col < 6):
            continue
        if abs(col - maze.c // 2) < 3:
            continue
        if maze.grid[row][col] == E:
            maze.grid[row][col] = F
            maze.grid[row][maze.c - col - 1] = F
            total_food += 2


MAX_DIFFERENT_MAZES = 10000


def generate_maze(seed=None, rows=18, cols=9, use_global=False):
    """mazeGenerator.generateMaze (:255-264).  rows x cols is the half maze before mirroring and bordering; the
    reference hard-codes Maze(18, 9), i.e. a 20 x 20 map.  Other sizes are a build-side extension with no reference
    counterpart (BASELINE config 5 names 32x16: rows=14, cols=15)."""
    if use_global:
        if not seed:
            seed = _random.randint(1, MAX_DIFFERENT_MAZES)
        _random.seed(seed)
        rng = _random
    else:
        if not seed:
            seed = _random.randint(1, MAX_DIFFERENT_MAZES)
        rng = _random.Random(seed)
    maze = _Maze(rows, cols, rng=rng)
    gapfactor = rng.gauss(0.7, 0.2)
    skip = _make_with_prison(maze, depth=0, gaps=5, vert=True, min_width=0, gapfactor=gapfactor)
    maze.to_map()
    _add_pacman_stuff(maze, 2 * (maze.r * maze.c // 20), 2, skip)
    return str(maze)


def random_layout(seed=None, use_global=True):
    """capture.randomLayout (capture.py:905-911): seed drawn from the global stream when not given."""
    if not seed:
        seed = _random.randint(0, 99999999)
    return generate_maze(seed, use_global=use_global)
