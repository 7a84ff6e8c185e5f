"""Synthetic community for the end-to-end validation of `profile --genome` (SURVEY.md section 8f-4,
BASELINE.json configs[4]).  The reference's Tier-2 validation (validation/generate_synthetic_alignments.py)
needs real genomes from NCBI; this generator keeps its SAM *model* on synthetic chromosomes:

* paired-end 2 x 100 bp reads from 175-bp inserts, CIGAR 100M, 0-3 substitutions per mate,
  NM / MD / AS = 100 - 2 NM tags, records grouped by QNAME, @HD SO:queryname;
* source genome ~ cell abundance x chromosome length;
* multi-mappers are the genuine exact occurrences of the 175-bp insert in any chromosome, either
  strand (shared loci are planted for that); every occurrence is written as a mate pair, the source
  pair primary, the others secondary;
* a fraction of inserts has one mapped mate only (one record, mate-unmapped flag).

Truth: per-insert source genome, the cell abundances, and how many inserts hit more than one
feature.  Written from scratch for this repository; only the model above follows the reference."""
from collections import defaultdict

import numpy as np

READ, INSERT = 100, 175
COMP = str.maketrans("ACGT", "TGCA")


def revcomp(s):
    return s.translate(COMP)[::-1]


def md_nm(ref, query):
    """MD:Z and NM for a gap-free alignment (substitutions only)."""
    out, run, nm = [], 0, 0
    for r, q in zip(ref, query):
        if r == q:
            run += 1
        else:
            out.append(str(run))
            out.append(r)
            run = 0
            nm += 1
    out.append(str(run))
    return "".join(out), nm


def build_flag(mate, reverse, mate_reverse, secondary, mate_present):
    """FLAG of one record: paired, READ1/READ2, strand, proper pair + mate strand when the mate is
    mapped (mate-unmapped otherwise), secondary for alternative occurrences."""
    flag = 0x1 | (0x40 if mate == 1 else 0x80)
    if reverse:
        flag |= 0x10
    if mate_present:
        flag |= 0x2
        if mate_reverse:
            flag |= 0x20
    else:
        flag |= 0x8
    if secondary:
        flag |= 0x100
    return flag


def occurrence_geometry(start, orientation, mate):
    """(start0, reverse, mate_start0, mate_reverse, tlen) of one mate at a 175-bp occurrence: the mate on
    the forward strand (mate 1 of a '+' occurrence, mate 2 of a '-' one) is leftmost."""
    fwd_mate = 1 if orientation == "+" else 2
    if mate == fwd_mate:
        return start, False, start + INSERT - READ, True, INSERT
    return start + INSERT - READ, True, start, False, -INSERT


def sam_record(qname, chrom_name, chrom_seq, start, orientation, mate, both_mapped, is_source, read):
    """One SAM line (no newline).  read = the mate's sequence as sequenced; the record carries it in
    reference orientation.  AS = 100 - 2 NM; QUAL is all 'I'; MAPQ 255."""
    pos0, reverse, mpos0, mate_reverse, tlen = occurrence_geometry(start, orientation, mate)
    seq = revcomp(read) if reverse else read
    md, nm = md_nm(chrom_seq[pos0:pos0 + READ], seq)
    if both_mapped:
        rnext, pnext = "=", mpos0 + 1
    else:
        rnext, pnext, tlen, mate_reverse = "*", 0, 0, False
    flag = build_flag(mate, reverse, mate_reverse, not is_source, both_mapped)
    return "\t".join([qname, str(flag), chrom_name, str(pos0 + 1), "255", f"{READ}M", rnext, str(pnext), str(tlen), seq,
                      "I" * READ, f"NM:i:{nm}", f"MD:Z:{md}", f"AS:i:{READ - 2 * nm}"])


def largest_remainder_counts(probabilities, total):
    """Integer allocation of `total` by probability: floors, then the largest remainders (ties by key)."""
    exact = {k: probabilities[k] * total for k in probabilities}
    counts = {k: int(np.floor(v)) for k, v in exact.items()}
    order = sorted(probabilities, key=lambda k: (-(exact[k] - counts[k]), k))
    for k in order[:total - sum(counts.values())]:
        counts[k] += 1
    return counts


def parse_profile_text(text):
    """(feature names, values, {total, mapped, multi-mapped inserts}) of a pandas-style profile."""
    import re
    head = "\n".join(line for line in text.split("\n") if line.startswith("#"))
    rows = [line.split("\t") for line in text.split("\n") if line.strip() and not line.startswith("#")]
    assert rows and rows[0][0] == "ID"

    def get(name):
        return int(re.search(name + r"\s*:\s*([0-9]+)", head).group(1))
    counts = {"reported_total_inserts": get("Total inserts"), "reported_mapped_inserts": get("Mapped inserts"),
              "reported_multimapped_inserts": get("Multiple mapped")}
    return [r[0] for r in rows[1:]], [float(r[1]) for r in rows[1:]], counts


class Community:
    def __init__(self, seed=1, n_species=3, strains=(3, 2, 2), chrom_len=(14000, 22000), n_shared=36, n_within=4,
                 sharing=True, copies=(1, 3)):
        rng = np.random.RandomState(seed)
        self.rng = rng
        self.genomes = []      # dict(name, species, strain, seq)
        for sp in range(n_species):
            for st in range(strains[sp]):
                L = int(rng.randint(chrom_len[0], chrom_len[1]))
                seq = "".join(rng.choice(list("ACGT"), size=L))
                self.genomes.append(dict(name=f"NC_{sp}{st}{rng.randint(1000, 9999)}.1", species=f"Species_{sp}",
                                         strain=f"Species_{sp}_strain_{st}", seq=seq))
        self.shared_sites = []  # (genome, start) of planted copies, to draw enriched inserts from
        if sharing:
            for k in range(n_shared + n_within):
                donor = int(rng.randint(len(self.genomes)))
                L = int(rng.randint(INSERT, 420))
                ds = int(rng.randint(0, len(self.genomes[donor]["seq"]) - L))
                piece = self.genomes[donor]["seq"][ds:ds + L]
                within = k >= n_shared
                others = [donor] if within else [g for g in range(len(self.genomes)) if g != donor]
                for _ in range(int(rng.randint(copies[0], copies[1]))):
                    g = int(rng.choice(others))
                    s = int(rng.randint(0, len(self.genomes[g]["seq"]) - L))
                    if within and abs(s - ds) < L:
                        continue
                    ins = revcomp(piece) if rng.rand() < 0.3 else piece
                    sq = self.genomes[g]["seq"]
                    self.genomes[g]["seq"] = sq[:s] + ins + sq[s + L:]
                    self.shared_sites.append((g, s, L))
                self.shared_sites.append((donor, ds, L))
        ab = rng.gamma(2.0, 1.0, size=len(self.genomes)) + 0.05
        self.abundance = ab / ab.sum()
        # exact 175-mer index, both strands: kmer -> [(genome, start, orient)]
        self.index = defaultdict(list)
        for gi, g in enumerate(self.genomes):
            s = g["seq"]
            for p in range(len(s) - INSERT + 1):
                self.index[s[p:p + INSERT]].append((gi, p, "+"))
        rc_hits = defaultdict(list)
        for gi, g in enumerate(self.genomes):
            r = revcomp(g["seq"])
            n = len(r)
            for p in range(n - INSERT + 1):
                k = r[p:p + INSERT]
                if k in self.index:
                    rc_hits[k].append((gi, n - INSERT - p, "-"))
        for k, v in rc_hits.items():
            self.index[k].extend(v)

    def genome_map(self, level):
        """lines 'feature<TAB>reference' of the --genome definition (level: 'strain' or 'species')."""
        return "".join(f"{g[level]}\t{g['name']}\n" for g in self.genomes)

    def feature_of(self, level):
        return [g[level] for g in self.genomes]

    def write(self, path, n_inserts, shared_fraction=0.25, single_mate_fraction=0.06, seed=2, exclude_cross=False,
              weighted_shared=False):
        """Writes the SAM; returns dict(source=[genome index per insert], targets=[set of genome indices]).
        weighted_shared: the inserts drawn from shared loci take their source genome in proportion to abundance x
        locus length as well (an absent genome is then never a source), instead of uniformly over the planted copies."""
        rng = np.random.RandomState(seed)
        lens = np.array([len(g["seq"]) for g in self.genomes], dtype=float)
        w = self.abundance * lens
        w /= w.sum()
        # inserts per source genome by largest remainder (as the reference allocates them), in random order
        alloc = largest_remainder_counts({i: float(w[i]) for i in range(len(w))}, n_inserts)
        plan = np.repeat(np.arange(len(w)), [alloc[i] for i in range(len(w))])
        rng.shuffle(plan)
        src, tgt = [], []
        site_p = None
        if weighted_shared and self.shared_sites:
            site_p = np.array([self.abundance[g] * L for g, _, L in self.shared_sites], dtype=float)
            site_p = site_p / site_p.sum() if site_p.sum() > 0 else None
        with open(path, "w") as f:
            f.write("@HD\tVN:1.6\tSO:queryname\n")
            for g in self.genomes:
                f.write(f"@SQ\tSN:{g['name']}\tLN:{len(g['seq'])}\n")
            i = 0
            while i < n_inserts:
                if self.shared_sites and rng.rand() < shared_fraction:
                    k_site = int(rng.choice(len(self.shared_sites), p=site_p)) if site_p is not None else int(rng.randint(len(self.shared_sites)))
                    gi, s0, L = self.shared_sites[k_site]
                    start = s0 + int(rng.randint(0, L - INSERT + 1))
                else:
                    gi = int(plan[i])
                    start = int(rng.randint(0, len(self.genomes[gi]["seq"]) - INSERT + 1))
                frag = self.genomes[gi]["seq"][start:start + INSERT]
                occ = list(self.index[frag])
                assert (gi, start, "+") in occ
                genomes_hit = {o[0] for o in occ}
                if exclude_cross and len(genomes_hit) > 1:
                    continue                                    # strict no-sharing control
                qname = f"ins{i:07d}"
                u = rng.rand()
                status = "both" if u >= single_mate_fraction else ("r1" if u < single_mate_fraction / 2 else "r2")
                reads = {}
                for mate, orig in ((1, frag[:READ]), (2, revcomp(frag[-READ:]))):
                    nm = int(rng.choice([0, 1, 2, 3], p=[0.5, 0.3, 0.1, 0.1]))
                    q = list(orig)
                    for p in rng.choice(READ, size=nm, replace=False):
                        q[p] = rng.choice([c for c in "ACGT" if c != q[p]])
                    reads[mate] = "".join(q)
                if status == "both":
                    todo = [(o, m) for o in occ for m in (1, 2)]
                else:
                    todo = [((gi, start, "+"), 1 if status == "r1" else 2)]
                    genomes_hit = {gi}
                for (g2, s2, orient), mate in todo:
                    f.write(sam_record(qname, self.genomes[g2]["name"], self.genomes[g2]["seq"], s2, orient, mate,
                                       status == "both", (g2, s2, orient) == (gi, start, "+"), reads[mate]) + "\n")
                src.append(gi)
                tgt.append(genomes_hit)
                i += 1
        return dict(source=src, targets=tgt)


def truth_rel(comm, level):
    """True relative cell abundance per feature at the given level."""
    out = defaultdict(float)
    for g, a in zip(comm.genomes, comm.abundance):
        out[g[level]] += float(a)
    return dict(out)


def bray_curtis(truth, est):
    keys = sorted(truth)
    l1 = sum(abs(truth[k] - est.get(k, 0.0)) for k in keys)
    den = sum(truth.values()) + sum(est.get(k, 0.0) for k in keys)
    return l1 / den if den > 0 else 0.0


def realised_rel(comm, truth, level):
    """Relative cell abundance per feature as the simulation realised it: source inserts per base of the feature."""
    cnt, length = defaultdict(float), defaultdict(float)
    for g in comm.genomes:
        length[g[level]] += len(g["seq"])
        cnt[g[level]] += 0.0
    for gi in truth["source"]:
        cnt[comm.genomes[gi][level]] += 1.0
    dens = {k: cnt[k] / length[k] for k in cnt}
    tot = sum(dens.values())
    return {k: v / tot for k, v in dens.items()}
