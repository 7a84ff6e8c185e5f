"""How close does the merge of identical multi-mapper lists come to the number of distinct feature sets?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, msamtools_amd as m, oracle_lib as orc
ctx = m.Context(0)
for G, R in ((200_000, 50), (200_000, 500), (1_000_000, 5000), (2_000_000, 100_000)):
    db = m.DeviceBatch.synth(ctx, 13579, G, R, 4)
    prof = m.Profile(ctx, R, "proportional")
    prof.accumulate(db, None)
    prof.finalize_enqueue(); prof.fetch()
    L0, E0 = prof.multi_size(); L, E = prof.shared_size()
    hs = m.HostSynth(13579, G, R, 4)
    goff = hs.group_off.astype(np.int64)
    sets = set()
    tid = hs.tid
    n_multi = 0
    for g in range(G):
        t = tid[goff[g]:goff[g + 1]]
        u = frozenset(t.tolist())
        if len(u) > 1:
            sets.add(u); n_multi += 1
    print(f"G={G} R={R}: lists {L0} (host {n_multi}), merged {L}, distinct sets {len(sets)}, merged/distinct {L/len(sets):.3f}, entries {E0}->{E}", flush=True)
    prof.close(); db.free()
