#!/usr/bin/env python3
"""Writes tests/golden/tokens_v1.mastok: 3 records (img 16, seg 4, text 8 tokens) from numpy RandomState(123), the byte-level
fixture of the token-shard format (make-a-scene_amd/token_data.py).  Nothing in the reference produces token files -- its
transformer loop only consumes them (train.py:141-145) -- so the fixture pins OUR format version 1, not a reference artefact."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "make-a-scene_amd"))
import token_data as TD  # noqa: E402

rs = np.random.RandomState(123)
img = rs.randint(0, 8192, (3, 16))
seg = rs.randint(0, 256, (3, 4))
text = rs.randint(1, 49664, (3, 8))
text[:, 4:] = 0
with TD.TokenShardWriter(os.path.join(HERE, "tokens_v1.mastok"), 16, 4, 8, 8192, 256, 49664) as w:
    w.append(img, seg, text)
print("wrote", os.path.join(HERE, "tokens_v1.mastok"))
