"""Cone classes of the CATER class table (reference object_indices.py:1-202): an object is a cone when its
name contains "_cone_" (is_cone_object, :200-202).  The table orders names size x colour x shape x material
with shape stride 4 inside the three 64-id size blocks, so the cone ids are every 4th id of 0..63 (large),
64..127 (medium)... - rather than carry the 193 names, the id set itself is stated here and pinned against the
reference's table by tests/test_datasets.py (fixture tests/golden/cone_ids.json)."""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "cone_ids.json")) as _f:
    CONE_IDS = frozenset(json.load(_f))

SNITCH_INDEX = 140            # reference baselines/datasets.py:14
NUM_CLASSES = 193


def is_cone_object(idx: int) -> int:
    return int(int(idx) in CONE_IDS)
