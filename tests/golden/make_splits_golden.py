#!/usr/bin/env python3
"""Writes tests/golden/splits_small.json: embeddings (C99 hex floats) and the splits oracle/pyref.py derives from them by
both algorithms of lib/Matrix.ml:524-612 (gaps: deterministic; centroids: with the declared SplitMix64 draws)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402

rng = np.random.RandomState(17)
cases = []
for n, d, keep in ((2, 1, 10), (6, 2, 4), (23, 5, 1000), (40, 3, 25)):
    centres = rng.normal(size=(3, d)) * 2
    emb = [[float(x) for x in centres[i % 3] + rng.normal(size=d) * 0.2] for i in range(n)]
    if n > 5:
        emb[4][0] = emb[1][0]       # equal coordinates in one dimension
        emb[5] = list(emb[2])       # two identical leaves
        emb[3][d - 1] = -0.0
    cases.append({"names": ["leaf %d" % i for i in range(n)], "keep": keep, "emb": [[x.hex() for x in row] for row in emb],
                  "gaps": [[g.hex(), m] for g, m in pyref.splits_gaps(emb, keep)],
                  "centroids": [[w.hex(), m] for w, m in pyref.splits_centroids(emb)]})
json.dump({"cases": cases}, open(os.path.join(HERE, "splits_small.json"), "w"), indent=0)
print("wrote", len(cases), "cases")
