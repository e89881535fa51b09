"""Loading helpers for the fixtures under tests/golden (data only; see golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SNAP_KEYS = ("pos", "dir", "pac", "scared", "carry", "ret", "food", "caps", "score")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta")).decode()) if "meta" in d else {}
    return d, meta


def names(pattern):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pattern)))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
