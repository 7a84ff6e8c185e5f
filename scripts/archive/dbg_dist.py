import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, msamtools_amd as m
G, R = int(sys.argv[1]) if len(sys.argv) > 1 else 60000, int(sys.argv[2]) if len(sys.argv) > 2 else 500

def run(order, path, fused=False):
    ctx = m.Context(0)
    if order == "before":
        ctx.dist_init(m.dist_unique_id(), 0, 1)
    db = m.DeviceBatch.synth(ctx, 13579, G, R, 4)
    prof = m.Profile(ctx, R, "proportional")
    if fused:
        run_ = m.FilterRun(ctx, db, l=80, p=95, z=80, besthit=True)
        run_.enqueue_with_profile(prof); run_.finish()
    else:
        import msamtools_amd._lib as L
        prof.accumulate(db, None)
    if order == "after":
        ctx.dist_init(m.dist_unique_id(), 0, 1)
    if path == "dist":
        prof.finalize_dist_enqueue()
    else:
        prof.finalize_enqueue()
    ab, st = prof.fetch()
    res = (ab[:3].tolist(), st.iterations, st.delta[19])
    prof.close(); db.free(); ctx.close()
    return res

for fused in (False, True):
    for order in ("none", "after", "before"):
        for path in ("plain", "dist"):
            if order == "none" and path == "dist":
                continue
            print(fused, order, path, run(order, path, fused), flush=True)
