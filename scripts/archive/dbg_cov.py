#!/usr/bin/env python3
"""debug: msx_coverage_depths against the streamed path on a dense batch (few references)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msamtools_amd as m
ctx = m.Context(0)
ng, nr, tl = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (300000, 300, 3000)))
db = m.DeviceBatch.synth(ctx, 4242, ng, nr, 4)
a = np.concatenate(m.coverage(ctx, db, [tl] * nr))
b = np.concatenate(m.coverage(ctx, db, [tl] * nr, whole_sample=True))
bad = np.nonzero(a != b)[0]
print("records", db.n_records, "cells", a.size, "mismatches", bad.size)
if bad.size:
    tiles = np.unique(bad >> 13)
    print("tiles with mismatches", tiles[:40], "of", (a.size >> 13) + 1)
    for t in tiles[:6]:
        lo = t << 13
        w = np.nonzero(a[lo:lo + 8192] != b[lo:lo + 8192])[0]
        print(" tile", t, "first bad cell", w[0], "last", w[-1], "n", w.size, "streamed", a[lo + w[0]:lo + w[0] + 4], "whole", b[lo + w[0]:lo + w[0] + 4],
              "diff const?", np.unique(a[lo + w] - b[lo + w])[:8])
