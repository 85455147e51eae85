"""helpers for reading tests/golden (floats are stored as float.hex strings)"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def unhex(v):
    return np.array([float.fromhex(s) for s in v], dtype=np.float64)
