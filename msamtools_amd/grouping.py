"""Host-side pool boundaries (msx_batch.group_off) -- Python mirror of the
string rules the C host applies while decoding (msamtools_amd/csrc/host).

filter_pools : msam_filter.c:120-125,132-138,170 -- a pool closes when a
               record's QNAME differs from the QNAME of the last MAPPED record.
profile_pools: msam_profile.c:223-232 -- records with tid == -1 are skipped
               entirely; a pool closes when a QNAME differs from the previous
               non-skipped record's.
Both return uint32 group_off[n_groups+1] with group_off[0] == 0 and
group_off[-1] == n_records (n_groups == 0 for an empty batch).
"""
from __future__ import annotations

import numpy as np


def _names(rec):
    if getattr(rec, "qname_off", None) is not None:
        q = rec.qname.tobytes()
        off = rec.qname_off
        return [q[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    return list(np.asarray(rec.name_id).tolist())


def filter_pools(rec):
    n = int(rec.flag.shape[0])
    if n == 0:
        return np.zeros(1, np.uint32)
    names = _names(rec)
    unmapped = (np.asarray(rec.flag) & 4) != 0
    off = [0]
    prev = None
    for i in range(n):
        if prev is not None and names[i] != prev:
            off.append(i)
        if not unmapped[i]:
            prev = names[i]
    off.append(n)
    return np.asarray(off, dtype=np.uint32)


def profile_pools(rec):
    n = int(rec.flag.shape[0])
    if n == 0:
        return np.zeros(1, np.uint32)
    names = _names(rec)
    tid = np.asarray(rec.tid)
    off = [0]
    prev = None
    for i in range(n):
        if tid[i] == -1:
            continue
        if prev is not None and names[i] != prev:
            off.append(i)
        prev = names[i]
    off.append(n)
    return np.asarray(off, dtype=np.uint32)
