"""Random capture mazes on the host, bit-identical to what the reference's generator prints for a seed.

Behaviour pinned by tests/golden/mazes.json (G5, captured from mazeGenerator.generateMaze, mazeGenerator.py:255-264).
The reference draws from the GLOBAL `random` module after `random.seed(seed)`; a maze is therefore a pure function of
the seed provided the randint / random / gauss / choice / shuffle calls happen in the same sequence.  This module keeps
that sequence but not the reference's program: the half board is a flat byte array, rooms are plain
(top, left, height, width) records on an explicit stack (no Maze objects, no recursion), and the full map is assembled
once at the end.  Draws come from a private `random.Random(seed)` unless `use_global` asks for the reference's side
effect on the caller's stream ("RNG").

Reference quirks that shape the output and are kept on purpose:
  * the "does this room touch the far border" test of a vertical wall compares ROW indices with the half board's
    column count (and the horizontal one compares columns with the row count), mazeGenerator.py:96,110;
  * the prison draw (randint, random) is consumed although no prison is ever built (:124-134);
  * capsule columns are tested against the float centre c/2, food columns against the integer centre c//2 (:232,:246).
"""
import random as _random

WALL, OPEN, PELLET, POWER = 0x25, 0x20, 0x2E, 0x6F          # '%', ' ', '.', 'o'
MAX_DIFFERENT_MAZES = 10000


class _Half:
    """The left half before mirroring: `cell[r * cols + c]`, row 0 at the top."""

    __slots__ = ("rows", "cols", "cell")

    def __init__(self, rows, cols):
        self.rows, self.cols = rows, cols
        self.cell = bytearray([OPEN]) * (rows * cols)

    def is_open(self, r, c):
        return self.cell[r * self.cols + c] == OPEN

    def block(self, r, c):
        self.cell[r * self.cols + c] = WALL


def _split(half, rng, room, at, gaps, vertical):
    """Draw one wall through `room` at offset `at`, leaving `gaps` openings; returns the two sub-rooms or None when the
    wall would seal the room off (mazeGenerator.py:85-118 as behaviour: which cells end up walls, and when shuffle runs)."""
    top, left, height, width = room
    if vertical:
        span, first, line = height, top, left + at
        probe = lambda k: half.is_open(k, line)                  # cell of the wall's own column in row k
        far_edge, limit = half.cols - 1, half.rows               # (sic) rows tested against the column count
    else:
        span, first, line = width, left, top + at
        probe = lambda k: half.is_open(line, k)
        far_edge, limit = half.rows - 1, half.cols
    gaps = min(span, gaps)
    run = list(range(first, first + span))
    # a wall must not plug the opening of the wall it abuts: drop its end cell when the cell beyond that end is open
    if first != 0:
        if probe(first - 1):
            run.pop(0)
        if len(run) <= gaps:
            return None
    last = first + span - 1
    if far_edge not in run and last + 1 < limit:
        if probe(last + 1):
            run.pop()
    if len(run) <= gaps:
        return None
    rng.shuffle(run)
    for k in run[int(round(gaps)):]:
        if vertical:
            half.block(k, line)
        else:
            half.block(line, k)
    if vertical:
        return (top, left, height, at), (top, left + at + 1, height, width - at - 1)
    return (top, left, at, width), (top + at + 1, left, height - at - 1, width)


def _carve(half, rng, gaps, gapfactor):
    """Recursive division, depth first and first child first, as an explicit stack of (room, gaps, vertical)."""
    todo = [((0, 0, half.rows, half.cols), gaps, True)]
    while todo:
        room, g, vertical = todo.pop()
        _, _, height, width = room
        if height <= 0 and width <= 0:
            continue
        extent = width if vertical else height
        if extent < 2:                                           # too thin this way: cut it the other way
            vertical = not vertical
            extent = width if vertical else height
        if extent <= 2:                                          # no interior line to put a wall on
            continue
        at = rng.choice(range(1, extent - 1))
        parts = _split(half, rng, room, at, g, vertical)
        if parts is None:
            continue
        g2 = max(1, g * gapfactor)
        todo.append((parts[1], g2, not vertical))                # popped second
        todo.append((parts[0], g2, not vertical))


def _full_map(half):
    """Mirror the half to the right and put a wall frame around it -> list of bytearray rows."""
    width = 2 * half.cols + 2
    rows = [bytearray([WALL]) * width]
    for r in range(half.rows):
        line = half.cell[r * half.cols:(r + 1) * half.cols]
        rows.append(bytearray([WALL]) + line + line[::-1] + bytearray([WALL]))
    rows.append(bytearray([WALL]) * width)
    return rows


def _dead_end_pellets(grid, skip):
    """Up to two sweeps that put a pellet (and its mirror image) into every open left-half cell with exactly one open
    neighbour; each sweep looks at the board as it was when the sweep began.  Returns the number of pellets placed."""
    n_rows, n_cols = len(grid), len(grid[0])
    placed = 0
    for _ in range(2):
        before = [bytes(r) for r in grid]
        added = 0
        for r in range(1, n_rows - 1):
            for c in range(1 + skip, n_cols // 2 - 1):
                if (r > n_rows - 6 and c < 6) or before[r][c] != OPEN:
                    continue
                exits = ((before[r - 1][c] == OPEN) + (before[r + 1][c] == OPEN) + (before[r][c - 1] == OPEN)
                         + (before[r][c + 1] == OPEN))
                if exits == 1:
                    grid[r][c] = grid[r][n_cols - 1 - c] = PELLET
                    added += 2
        placed += added
        if not added:
            break
    return placed


def _scatter(grid, rng, glyph, want, have, col_hi, centre, skip, what):
    """Rejection-sample left-half cells (mirrored to the right) until `want` glyphs are on the board."""
    n_rows, n_cols = len(grid), len(grid[0])
    tries = 0
    while have < want:
        tries += 1
        if tries > 2000000:     # the reference spins forever when nothing qualifies (only possible for non-reference sizes)
            raise ValueError(f"maze size leaves too few legal {what} cells")
        r = rng.randint(1, n_rows - 1)
        c = rng.randint(1 + skip, col_hi)
        if (r > n_rows - 6 and c < 6) or abs(c - centre) < 3:
            continue
        if grid[r][c] == OPEN:
            grid[r][c] = grid[r][n_cols - 1 - c] = glyph
            have += 2
    return have


def generate_maze(seed=None, rows=18, cols=9, use_global=False):
    """The map text mazeGenerator.generateMaze(seed) returns.  rows x cols is the half board before mirroring and
    framing; the reference hard-codes 18 x 9 (a 20 x 20 map).  Other sizes are a build-side extension with no reference
    counterpart (BASELINE config 5 names 32x16: rows=14, cols=15)."""
    if not seed:
        seed = _random.randint(1, MAX_DIFFERENT_MAZES)
    if use_global:
        _random.seed(seed)
        rng = _random
    else:
        rng = _random.Random(seed)
    half = _Half(rows, cols)
    gapfactor = rng.gauss(0.7, 0.2)
    rng.randint(0, 2)           # the prison draws: consumed, never used (see the module docstring)
    rng.random()
    skip = 0
    _carve(half, rng, 5, gapfactor)
    grid = _full_map(half)
    n_rows, n_cols = len(grid), len(grid[0])
    pellets = _dead_end_pellets(grid, skip)
    for digit, (r, c) in zip(b"3142", ((1, 1), (2, 1), (1, n_cols - 2), (2, n_cols - 2))):
        grid[r][c] = digit
    _scatter(grid, rng, POWER, 2, 0, n_cols // 2 - 2, n_cols / 2, skip, "capsule")
    _scatter(grid, rng, PELLET, 2 * (n_rows * n_cols // 20), pellets, n_cols // 2 - 1, n_cols // 2, skip, "food")
    return "\n".join(r.decode("ascii") for r in grid)


def random_layout(seed=None, use_global=True):
    """capture.randomLayout (capture.py:905-911): seed drawn from the global stream when not given."""
    if not seed:
        seed = _random.randint(0, 99999999)
    return generate_maze(seed, use_global=use_global)
