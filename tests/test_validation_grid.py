"""The reference's Tier-2 validation design at its own scale (docs/validation/profile_validation.md:9-20,40-43;
validation/validate_profiles.py:131), on the GPU command line: insert counts 1 000 / 10 000 / 100 000 x RNG seeds
13579 / 24680 / 97531 x four communities (balanced_skew, similar_strains, rare_strain, absent_strain) x the four
--multi modes, plus the strict no-sharing controls.  The genomes are synthetic (tests/community.py: the reference's
SAM model on random chromosomes -- its real genomes need a download), the checks are the reference's:
  * header counts == truth (validate_profiles.py:859-876)
  * sum of relative abundances = 1 +- 5e-6 (:879)
  * exact per-reference insert recovery in the control, identical under every mode (:458-538)
  * mean Bray-Curtis distance to the generating composition orders prop < equal < all < ignore (the reference's
    table: 0.0021 < 0.0124 < 0.0281 < 0.0468), and an absent strain stays at exactly 0 under proportional sharing.
The whole grid runs by default: 36 simulations + 9 controls, as the reference's 45 (about three minutes, most of it
writing the 100 000-insert SAM files in Python).  MSX_VALIDATION_QUICK=1 gives the 100 000-insert size its first seed
only (round 3's default: 28 + 7).  A summary goes to $MSX_VALIDATION_OUT when set.
"""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import community as cm
from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)
SEEDS = (13579, 24680, 97531)
SIZES = (1000, 10000, 100000)
MODES = ("all", "equal", "ignore", "prop")


def make_community(kind, seed):
    """the four community designs on a 3-species / 7-strain synthetic reference set"""
    # every shared locus has one other copy (the reference's multi-mapping comes mostly from pairs of related strains);
    # similar_strains: two and a half times as many of them
    comm = cm.Community(seed=1000 + seed % 997, n_shared=90 if kind == "similar_strains" else 36, n_within=4, copies=(1, 2))
    rng = np.random.RandomState(seed)
    ab = rng.gamma(2.0, 1.0, size=len(comm.genomes)) + 0.05
    if kind == "balanced_skew":
        ab = np.sort(ab)[::-1] * np.array([4, 3, 2, 1.5, 1, 0.7, 0.5])[:len(ab)]
    elif kind == "rare_strain":
        ab[1] = 0.004 * ab.sum()                # a strain of species 0 at well under a percent
    elif kind == "absent_strain":
        ab[1] = 0.0                             # ... or not there at all, its sister strains abundant
    comm.abundance = ab / ab.sum()
    return comm


def run_modes(d, sam, gdef, n, unit="rel", nolen=False):
    out = {}
    for mode in MODES:
        p = os.path.join(d, f"p_{mode}.gz")
        args = [BIN, "profile", "-S", f"--unit={unit}", "--label", "v", "--genome", gdef, "--total", str(n), "--multi", mode, "-o", p]
        if nolen:
            args.append("--nolen")
        r = subprocess.run(args + [sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        names, vals, counts = cm.parse_profile_text(gzip.open(p, "rt").read())
        out[mode] = (names, np.array(vals), counts)
    return out


def test_validation_grid(tmp_path):
    full = os.environ.get("MSX_VALIDATION_QUICK") != "1"
    d = str(tmp_path)
    bc = {m: [] for m in MODES}
    absent_prop, rows = [], []
    n_main = n_ctrl = 0
    for n in SIZES:
        for seed in (SEEDS if (full or n < 100000) else SEEDS[:1]):
            for kind in ("balanced_skew", "similar_strains", "rare_strain", "absent_strain"):
                comm = make_community(kind, seed)
                sam, gdef = os.path.join(d, "a.sam"), os.path.join(d, "g.tsv")
                # main shared-locus target fraction 0.2; sources in proportion to abundance everywhere
                truth = comm.write(sam, n, shared_fraction=0.2, seed=seed, weighted_shared=True)
                open(gdef, "w").write(comm.genome_map("strain"))
                lv = comm.feature_of("strain")
                n_multi = sum(1 for t in truth["targets"] if len({lv[g] for g in t}) > 1)
                res = run_modes(d, sam, gdef, n)
                tr = cm.realised_rel(comm, truth, "strain")      # the composition the simulation realised (source inserts per base)
                for mode, (names, vals, counts) in res.items():
                    assert (counts["reported_total_inserts"], counts["reported_mapped_inserts"],
                            counts["reported_multimapped_inserts"]) == (n, n, n_multi), (kind, n, seed, mode)
                    assert abs(float(vals.sum()) - 1.0) <= 5e-6
                    est = dict(zip(names, [float(v) for v in vals]))
                    b = cm.bray_curtis(tr, est)
                    bc[mode].append(b)
                    rows.append(dict(community=kind, inserts=n, seed=seed, mode=mode, bray_curtis=b))
                    if kind == "absent_strain" and mode == "prop":
                        absent_prop.append(est[comm.genomes[1]["strain"]])
                n_main += 1
            # strict no-sharing control: exact recovery, every mode the same
            ctrl = cm.Community(seed=2000 + seed % 997, sharing=False)
            sam0, g0 = os.path.join(d, "c.sam"), os.path.join(d, "c.tsv")
            t0 = ctrl.write(sam0, n, shared_fraction=0.0, seed=seed + 1, exclude_cross=True)
            open(g0, "w").write(ctrl.genome_map("strain"))
            res = run_modes(d, sam0, g0, n, unit="ab", nolen=True)
            first = None
            for mode, (names, vals, counts) in res.items():
                exp = {nm: 0.0 for nm in names}
                for g in t0["source"]:
                    exp[ctrl.genomes[g]["strain"]] += 1
                assert [exp[nm] for nm in names] == list(vals), (n, seed, mode)        # maximum insert-count error: 0
                assert (counts["reported_total_inserts"], counts["reported_mapped_inserts"], counts["reported_multimapped_inserts"]) == (n, n, 0)
                first = vals if first is None else first
                assert (vals == first).all()
            n_ctrl += 1
    mean = {m: float(np.mean(v)) for m, v in bc.items()}
    mx = {m: float(np.max(v)) for m, v in bc.items()}
    summary = dict(main_simulations=n_main, controls=n_ctrl, profile_evaluations=len(rows), mean_bray_curtis=mean,
                   max_bray_curtis=mx, absent_strain_max_prop_abundance=max(absent_prop), full_grid=full)
    if os.environ.get("MSX_VALIDATION_OUT"):
        json.dump(dict(summary=summary, rows=rows), open(os.environ["MSX_VALIDATION_OUT"], "w"), indent=1)
    print(json.dumps(summary))
    assert mean["prop"] < mean["equal"] < mean["all"] < mean["ignore"], mean
    assert max(absent_prop) == 0.0            # no false-positive abundance for the absent strain (proportional sharing)
    assert mean["prop"] < 0.02, mean
