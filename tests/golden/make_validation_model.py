#!/usr/bin/env python3
"""Writes tests/golden/validation_model.json: input/output vectors of the reference's Tier-2
validation model, produced by IMPORTING validation/generate_synthetic_alignments.py and
validation/validate_profiles.py from /root/reference (build container only; the vectors are
data, the reference's sources never travel).  They pin tests/community.py -- the repository's
own generator -- and the checks of tests/test_validation_community.py to the reference's model:

  build_flag            all 32 argument combinations            (generate_synthetic_alignments.py:880-904)
  md_tag                reference/query pairs -> (MD, NM)       (:822-840)
  occurrence_geometry   start x orientation x mate              (:851-871)
  largest_remainder_counts                                      (:607-623)
  write_sam_and_truth   a tiny two-genome community: the SAM lines (layout :1043-1063, AS rule :1034)
                        and insert_sources.tsv the reference writes for given inserts
  validate_profiles     parse_profile_counts / read_profile on a profile text, calculate_metrics
                        (:379-420, :579-611; the checks at :859-879 compare exactly these numbers)

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_validation_model.py
"""
import gzip
import json
import os
import random
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
REF = "/root/reference/validation"
sys.path.insert(0, REF)
import generate_synthetic_alignments as gsa  # noqa: E402
import validate_profiles as vp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {"_source": "arumugamlab/msamtools validation/*.py imported in the build container; see this script"}

# ---- build_flag ---------------------------------------------------------------------------------------
out["build_flag"] = [
    {"mate": mate, "reverse": bool(rv), "mate_reverse": bool(mr), "secondary": bool(sec), "mate_present": bool(mp),
     "flag": gsa.build_flag(mate, bool(rv), bool(mr), bool(sec), bool(mp))}
    for mate in (1, 2) for rv in (0, 1) for mr in (0, 1) for sec in (0, 1) for mp in (0, 1)]

# ---- md_tag ---------------------------------------------------------------------------------------------
rng = random.Random(20251003)
pairs = []
for n_mm in (0, 1, 2, 3, 5, 100):
    for _ in range(3):
        ref = "".join(rng.choice("ACGT") for _ in range(100))
        q = gsa.mutate_read(ref, n_mm, rng)
        pairs.append((ref, q))
ref = "".join(rng.choice("ACGT") for _ in range(100))
for positions in ([0], [99], [0, 99], [10, 11], [0, 1, 2], [98, 99], [49, 50, 51, 52]):
    q = list(ref)
    for p in positions:
        q[p] = "A" if q[p] != "A" else "C"
    pairs.append((ref, "".join(q)))
out["md_tag"] = [{"reference": r, "query": q, "md": gsa.md_tag(r, q)[0], "nm": gsa.md_tag(r, q)[1]} for r, q in pairs]

# ---- occurrence_geometry --------------------------------------------------------------------------------
out["occurrence_geometry"] = [
    {"start": st, "orientation": o, "mate": m,
     "result": list(gsa.occurrence_geometry(gsa.Occurrence("asm", st, o), m))}
    for st in (0, 17, 1000) for o in "+-" for m in (1, 2)]

# ---- largest_remainder_counts ----------------------------------------------------------------------------
lrc = []
for probs, total in (({"a": 0.5, "b": 0.3, "c": 0.2}, 10), ({"a": 1 / 3, "b": 1 / 3, "c": 1 / 3}, 10),
                     ({"x": 0.105, "y": 0.105, "z": 0.79}, 19), ({"g1": 0.25, "g2": 0.25, "g3": 0.25, "g4": 0.25}, 7),
                     ({"only": 1.0}, 5), ({"a": 0.999, "b": 0.001}, 3)):
    lrc.append({"probabilities": probs, "total": total, "counts": gsa.largest_remainder_counts(probs, total)})
out["largest_remainder_counts"] = lrc

# ---- write_sam_and_truth on a tiny community ---------------------------------------------------------------
rng = random.Random(97531)
g1 = "".join(rng.choice("ACGT") for _ in range(700))
shared = g1[200:200 + 260]
g2 = "".join(rng.choice("ACGT") for _ in range(150)) + shared + "".join(rng.choice("ACGT") for _ in range(240))
g3 = "".join(rng.choice("ACGT") for _ in range(90)) + gsa.reverse_complement(shared) + "".join(rng.choice("ACGT") for _ in range(300))
genomes = [gsa.Genome("ASM_1", "Species a", "strain 1", "NC_000001.1", g1, Path("x1.fa")),
           gsa.Genome("ASM_2", "Species a", "strain 2", "NC_000002.1", g2, Path("x2.fa")),
           gsa.Genome("ASM_3", "Species b", "strain 3", "NC_000003.1", g3, Path("x3.fa"))]
by_asm = {g.assembly_accession: g for g in genomes}


def occurrences(asm, start):
    frag = by_asm[asm].sequence[start:start + gsa.INSERT_LENGTH]
    rc = gsa.reverse_complement(frag)
    occ = []
    for g in genomes:
        p = g.sequence.find(frag)
        while p >= 0:
            occ.append(gsa.Occurrence(g.assembly_accession, p, "+"))
            p = g.sequence.find(frag, p + 1)
        p = g.sequence.find(rc)
        while p >= 0:
            occ.append(gsa.Occurrence(g.assembly_accession, p, "-"))
            p = g.sequence.find(rc, p + 1)
    return occ


plan = [("ASM_1", 10, "both_mapped"), ("ASM_1", 230, "both_mapped"), ("ASM_2", 170, "both_mapped"),
        ("ASM_3", 100, "both_mapped"), ("ASM_2", 20, "r1_only"), ("ASM_3", 350, "r2_only"),
        ("ASM_1", 250, "r1_only"), ("ASM_3", 120, "both_mapped")]
selected = []
for i, (asm, start, status) in enumerate(plan):
    occ = occurrences(asm, start)
    assert gsa.Occurrence(asm, start, "+") in occ
    selected.append(gsa.SelectedInsert(qname=f"sim{i:08d}", source_assembly=asm, source_start=start, fragment_code=0,
                                       occurrences=occ, selection_mode="fixture", mate_status=status))
counts, weights = gsa.parse_mismatch_weights(gsa.DEFAULT_MISMATCH_WEIGHTS)
with tempfile.TemporaryDirectory() as td:
    gsa.write_sam_and_truth(Path(td), genomes, selected, by_asm, counts, weights, random.Random(13579))
    sam_lines = open(os.path.join(td, "alignments.sam")).read().split("\n")
    sources = open(os.path.join(td, "insert_sources.tsv")).read()
out["sam_model"] = {
    "genomes": [{"assembly": g.assembly_accession, "chromosome": g.chromosome_accession, "sequence": g.sequence} for g in genomes],
    "inserts": [{"qname": s.qname, "source_assembly": s.source_assembly, "source_start": s.source_start,
                 "mate_status": s.mate_status, "mate1_nm": s.mate1_nm, "mate2_nm": s.mate2_nm,
                 "occurrences": [[o.assembly_accession, o.start, o.orientation] for o in s.occurrences]} for s in selected],
    "sam_lines": [l for l in sam_lines if l],
    "insert_sources_tsv": sources,
}

# ---- validate_profiles: what the checks at :859-879 read from a profile ----------------------------------
profile_text = """# msamtools version: 1.1.3
# Command: msamtools profile --multi=proportional --label=test --genome=g.tsv --total=3000 -o out.gz in.sam
#   Total inserts          :       3000
#   Mapped inserts         :       2987 ( 99.57%)
#     - Multiple mapped    :        431 ( 14.43%)
#     - Uniquely mapped    :       2556 ( 85.57%)
ID\ttest
Unknown\t0.0043333333
strain 1\t0.41
strain 2\t0.3256666667
strain 3\t0.26
"""
with tempfile.TemporaryDirectory() as td:
    pth = Path(td) / "p.tsv.gz"
    with gzip.open(pth, "wt") as fh:
        fh.write(profile_text)
    out["profile_parse"] = {"text": profile_text, "counts": vp.parse_profile_counts(pth), "values": vp.read_profile(pth)}
truth = {"Unknown": 0.0, "strain 1": 0.4, "strain 2": 0.35, "strain 3": 0.25}
est = out["profile_parse"]["values"]
out["calculate_metrics"] = [{"truth": truth, "estimate": est, "metrics": vp.calculate_metrics(truth, est)},
                            {"truth": {"a": 0.5, "b": 0.5}, "estimate": {"a": 0.5, "b": 0.5},
                             "metrics": {k: v for k, v in vp.calculate_metrics({"a": 0.5, "b": 0.5, "c": 0.0},
                                                                               {"a": 0.5, "b": 0.5, "c": 0.0}).items()
                                         if k in ("mae", "rmse", "tvd", "bray_curtis", "max_abs_error")},
                             "features_note": "computed over a/b/c with c = 0 in both"}]
out["sum_tolerance"] = 5e-6          # validate_profiles.py:879

with open(os.path.join(HERE, "validation_model.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print("wrote validation_model.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})
