"""Static map handling on the host: the reference's layout.py restated for bit-board rows.

  Layout.from_text / from_file   layout.py:27-37, 95-130 (processLayoutText / processLayoutChar) and
                                 layout.py:131-149 (getLayout path search)
Coordinates are (x, y) with the origin bottom-left; the text's first row is the TOP row (layout.py:108-112).
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LAYOUT_DIR = os.path.join(_HERE, "layouts")   # the reference's three maps (layouts/*.lay), data files


class Layout:
    def __init__(self, rows):
        rows = [r for r in rows]
        if not rows or any(len(r) != len(rows[0]) for r in rows):
            raise ValueError("layout rows must be non-empty and of equal length")
        self.text = rows
        self.height, self.width = len(rows), len(rows[0])
        if self.width > 32 or self.height > 32:
            raise ValueError("layouts wider or taller than 32 cells are not supported (one uint32 per row)")
        H, W = self.height, self.width
        self.wall_rows = np.zeros(H, np.uint32)
        self.food_rows = np.zeros(H, np.uint32)
        self.cap_rows = np.zeros(H, np.uint32)
        self.capsules = []
        agents = []
        for y in range(H):
            for x in range(W):
                ch = rows[H - 1 - y][x]
                if ch == "%":
                    self.wall_rows[y] |= np.uint32(1 << x)
                elif ch == ".":
                    self.food_rows[y] |= np.uint32(1 << x)
                elif ch == "o":
                    self.cap_rows[y] |= np.uint32(1 << x)
                    self.capsules.append((x, y))
                elif ch in ("1", "2", "3", "4"):
                    agents.append((int(ch), (x, y)))
                # 'P'/'G' are classic-Pacman markers: capture layouts use the digits (layout.py:123-130)
        agents.sort()                                   # layout.py:113
        if [d for d, _ in agents] != [1, 2, 3, 4]:
            raise ValueError("a capture layout needs exactly the agent digits 1, 2, 3, 4")
        self.agent_positions = [p for _, p in agents]   # agent index = digit - 1
        self.starts = np.array(self.agent_positions, np.int8)
        self.total_food = int(sum(bin(int(r)).count("1") for r in self.food_rows))   # layout.py:37

    @classmethod
    def from_text(cls, text):
        if isinstance(text, str):
            text = text.split("\n")
        return cls([ln.strip() for ln in text if ln.strip() != ""])

    @classmethod
    def from_file(cls, path):
        with open(path) as f:
            return cls([ln.strip() for ln in f if ln.strip() != ""])     # layout.py:145-149 tryToLoad

    def is_wall(self, x, y):
        return bool((int(self.wall_rows[y]) >> x) & 1)

    def open_cells(self):
        """Grid.asList(False) order: x outer, y inner (game.py:225-230)."""
        return [(x, y) for x in range(self.width) for y in range(self.height) if not self.is_wall(x, y)]

    def __str__(self):
        return "\n".join(self.text)


def find_layout_file(name):
    """layout.getLayout's search, minus the chdir walk: as given, under layouts/, with and without '.lay'."""
    cands = [name, name + ".lay", os.path.join("layouts", name), os.path.join("layouts", name + ".lay"),
             os.path.join(LAYOUT_DIR, os.path.basename(name)), os.path.join(LAYOUT_DIR, os.path.basename(name) + ".lay")]
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def get_layout(name):
    p = find_layout_file(name)
    return Layout.from_file(p) if p else None
