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


class Community:
    def __init__(self, seed=1, n_species=3, strains=(3, 2, 2), chrom_len=(14000, 22000), n_shared=36, n_within=4,
                 sharing=True):
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
                for _ in range(int(rng.randint(1, 3))):
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

    def write(self, path, n_inserts, shared_fraction=0.25, single_mate_fraction=0.06, seed=2, exclude_cross=False):
        """Writes the SAM; returns dict(source=[genome index per insert], targets=[set of genome indices])."""
        rng = np.random.RandomState(seed)
        lens = np.array([len(g["seq"]) for g in self.genomes], dtype=float)
        w = self.abundance * lens
        w /= w.sum()
        src, tgt = [], []
        with open(path, "w") as f:
            f.write("@HD\tVN:1.6\tSO:queryname\n")
            for g in self.genomes:
                f.write(f"@SQ\tSN:{g['name']}\tLN:{len(g['seq'])}\n")
            i = 0
            while i < n_inserts:
                if self.shared_sites and rng.rand() < shared_fraction:
                    gi, s0, L = self.shared_sites[int(rng.randint(len(self.shared_sites)))]
                    start = s0 + int(rng.randint(0, L - INSERT + 1))
                else:
                    gi = int(rng.choice(len(self.genomes), p=w))
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
                    fwd_mate = 1 if orient == "+" else 2            # the mate lying on the forward strand at s2
                    reverse = mate != fwd_mate
                    pos0 = s2 if not reverse else s2 + INSERT - READ
                    mpos0 = s2 + INSERT - READ if not reverse else s2
                    seq = revcomp(reads[mate]) if reverse else reads[mate]
                    ref = self.genomes[g2]["seq"][pos0:pos0 + READ]
                    md, nm = md_nm(ref, seq)
                    flag = 0x1 | (0x40 if mate == 1 else 0x80) | (0x10 if reverse else 0)
                    if status == "both":
                        flag |= 0x2 | (0x20 if not reverse else 0)
                        rnext, pnext, tlen = "=", mpos0 + 1, (INSERT if not reverse else -INSERT)
                    else:
                        flag |= 0x8
                        rnext, pnext, tlen = "*", 0, 0
                    if (g2, s2, orient) != (gi, start, "+"):
                        flag |= 0x100
                    f.write("\t".join([qname, str(flag), self.genomes[g2]["name"], str(pos0 + 1), "255", f"{READ}M", rnext,
                                       str(pnext), str(tlen), seq, "I" * READ, f"NM:i:{nm}", f"MD:Z:{md}",
                                       f"AS:i:{READ - 2 * nm}"]) + "\n")
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
