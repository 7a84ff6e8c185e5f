"""tests/community.py -- the repository's own generator of the Tier-2 validation community -- and the
profile checks of test_validation_community.py, pinned to the reference's model through vectors produced
by the reference's own validation/*.py (tests/golden/make_validation_model.py wrote
tests/golden/validation_model.json in the build container)."""
import json
import os

import pytest

import community as cm
from conftest import GOLDEN

M = json.load(open(os.path.join(GOLDEN, "validation_model.json")))


def test_build_flag_table():
    for c in M["build_flag"]:        # generate_synthetic_alignments.py:880-904, all 32 cases
        assert cm.build_flag(c["mate"], c["reverse"], c["mate_reverse"], c["secondary"], c["mate_present"]) == c["flag"], c


def test_md_tag_and_as_rule():
    for c in M["md_tag"]:            # :822-840
        assert cm.md_nm(c["reference"], c["query"]) == (c["md"], c["nm"])


def test_occurrence_geometry():
    for c in M["occurrence_geometry"]:   # :851-871
        assert list(cm.occurrence_geometry(c["start"], c["orientation"], c["mate"])) == c["result"], c


def test_largest_remainder_counts():
    for c in M["largest_remainder_counts"]:   # :607-623
        assert cm.largest_remainder_counts(c["probabilities"], c["total"]) == c["counts"]


def test_sam_lines_of_a_tiny_community():
    """Every record line write_sam_and_truth produced (layout :1043-1063, flags, geometry, NM/MD/AS) is
    reproduced from (genome, occurrence, mate, mate status, source?, read as sequenced)."""
    sm = M["sam_model"]
    chrom = {g["assembly"]: (g["chromosome"], g["sequence"]) for g in sm["genomes"]}
    lines = sm["sam_lines"]
    assert lines[0] == "@HD\tVN:1.6\tSO:queryname"
    assert lines[1:1 + len(sm["genomes"])] == [f"@SQ\tSN:{g['chromosome']}\tLN:{len(g['sequence'])}" for g in sm["genomes"]]
    recs = [l.split("\t") for l in lines if not l.startswith("@")]
    by_q = {}
    for r in recs:
        by_q.setdefault(r[0], []).append(r)
    assert list(by_q) == [i["qname"] for i in sm["inserts"]]          # QNAME-grouped, in insert order
    n_checked = 0
    for ins in sm["inserts"]:
        got = by_q[ins["qname"]]
        both = ins["mate_status"] == "both_mapped"
        if both:
            todo = [(tuple(o), m) for o in ins["occurrences"] for m in (1, 2)]
        else:
            todo = [((ins["source_assembly"], ins["source_start"], "+"), 1 if ins["mate_status"] == "r1_only" else 2)]
        assert len(got) == len(todo)
        # the reads as sequenced, recovered from the source occurrence's records
        reads = {}
        for r, ((asm, st, orient), mate) in zip(got, todo):
            rev = bool(int(r[1]) & 0x10)
            reads.setdefault(mate, cm.revcomp(r[9]) if rev else r[9])
        for r, ((asm, st, orient), mate) in zip(got, todo):
            name, seq = chrom[asm]
            is_source = (asm, st, orient) == (ins["source_assembly"], ins["source_start"], "+")
            want = cm.sam_record(ins["qname"], name, seq, st, orient, mate, both, is_source, reads[mate])
            assert want == "\t".join(r), (ins["qname"], asm, st, orient, mate)
            assert int(r[11].split(":")[2]) == (ins["mate1_nm"] if mate == 1 else ins["mate2_nm"])
            n_checked += 1
    assert n_checked == len(recs) and n_checked >= 20
    # insert_sources.tsv: qname, source, mate status
    rows = [l.split("\t") for l in sm["insert_sources_tsv"].strip().split("\n")]
    assert rows[0] == ["qname", "source_assembly", "mate_status"]
    assert [tuple(r) for r in rows[1:]] == [(i["qname"], i["source_assembly"], i["mate_status"]) for i in sm["inserts"]]


def test_profile_parsing_and_metrics():
    pp = M["profile_parse"]           # validate_profiles.py:379-420
    names, values, counts = cm.parse_profile_text(pp["text"])
    assert counts == pp["counts"]
    assert dict(zip(names, values)) == pp["values"]
    assert abs(sum(values) - 1.0) <= M["sum_tolerance"]                # :879
    for c in M["calculate_metrics"][:1]:                               # :579-611
        assert cm.bray_curtis(c["truth"], c["estimate"]) == pytest.approx(c["metrics"]["bray_curtis"], rel=1e-12)
