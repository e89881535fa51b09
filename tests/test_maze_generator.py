"""CPU suite: the host maze generator against G5 (mazeGenerator.generateMaze output captured from the reference)."""
import random

import _golden as G


def test_mazes_match_reference():
    from pmx import maze_generator as MG
    from pmx.layout import Layout
    g = G.load_json("mazes.json")["mazes"]
    assert len(g) >= 20
    for key, rows in g.items():
        seed = 7 if key.startswith("randomLayout") else int(key)
        assert MG.generate_maze(seed).split("\n") == rows, f"seed {seed}"
        lay = Layout.from_text(rows)
        assert (lay.width, lay.height) == (20, 20)
        assert lay.agent_positions == [(1, 17), (18, 17), (1, 18), (18, 18)]
        assert len(lay.capsules) == 2


def test_global_stream_side_effect_is_opt_in():
    from pmx import maze_generator as MG
    random.seed(5)
    a = random.random()
    random.seed(5)
    MG.generate_maze(23)                       # private generator: the global stream is untouched
    assert random.random() == a
    MG.generate_maze(23, use_global=True)      # the reference's behaviour: reseeds the global module
    b = random.random()
    random.seed(23)
    MG.generate_maze(23, use_global=True)
    assert random.random() == b


def test_other_sizes_are_valid_layouts():
    from pmx import maze_generator as MG
    from pmx.layout import Layout
    lay = Layout.from_text(MG.generate_maze(3, rows=14, cols=15))     # 32 x 16 (BASELINE config 5 wording)
    assert (lay.width, lay.height) == (32, 16)
    assert len(lay.agent_positions) == 4
