"""CPU-side checks of the drop-in boundary: the shared object loads, exports
every symbol include/msamtools_amd.h declares, and refuses to run without a
gfx950 GPU (no fallback path)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "msamtools_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from msamtools_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msamtools_amd.h but not exported"
    # and the binding table covers the header exactly
    assert sorted(_lib.SYMBOLS) == names
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (msx_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_abi_version():
    from msamtools_amd import _lib
    assert _lib.load().msx_abi_version() == 7


def test_library_carries_gfx950_code_object():
    from msamtools_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    for kern in (b"k_aln_stats_flat", b"k_besthit_select", b"k_insert_count", b"k_share_reduce", b"k_rs_scatter"):
        assert kern in data


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import msamtools_amd as m
    with pytest.raises(m.MsxError) as ei:
        m.Context(0)
    assert ei.value.code == m._lib.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under msamtools_amd/ or include/ may reference it."""
    bad = []
    for base in ("msamtools_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".c", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"msx_oracle|oracle_lib|libmsx_oracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_filter_params_validation():
    import msamtools_amd as m
    fp = m.filter_params(l=80, p=95, z=80, besthit=True)
    assert (fp.min_length, fp.ppt, fp.max_clip, fp.besthit) == (80, 950, 20, 1)
    for bad in (dict(p=101), dict(ppt=1001), dict(z=101), dict(l=-1), dict(p=1, ppt=1),
                dict(besthit=True, uniqhit=True), dict(invert=True, besthit=True)):
        with pytest.raises(ValueError):
            m.filter_params(**bad)


def test_host_synth_is_deterministic_and_prefix_stable():
    import msamtools_amd as m
    a = m.HostSynth(13579, 3000, 500, 4)
    b = m.HostSynth(13579, 3000, 500, 4)
    assert a.n_records == b.n_records and (a.md == b.md).all() and (a.cigar == b.cigar).all()
    # groups [1000, 2000) generated on their own equal the middle of the big batch
    c = m.HostSynth(13579, 1000, 500, 4, first_group=1000)
    s, e = int(a.group_off[1000]), int(a.group_off[2000])
    assert c.n_records == e - s
    assert (c.flag == a.flag[s:e]).all() and (c.tid == a.tid[s:e]).all() and (c.as_ == a.as_[s:e]).all()
    ms, me = int(a.md_off[s]), int(a.md_off[e])
    assert (c.md == a.md[ms:me]).all()
    other = m.HostSynth(24680, 3000, 500, 4)
    assert other.n_records != a.n_records or not (other.tid == a.tid).all()


def test_host_synth_model_is_self_consistent():
    """MD/CIGAR/NM/AS of the synthetic stream agree with each other (checked with the oracle)."""
    import msamtools_amd as m
    import oracle_lib as orc
    hs = m.HostSynth(97531, 20000, 1000, 4)
    st = orc.aln_stats(hs)
    assert (st["edit"] == hs.nm).all() and (st["qlen"] == 100).all()
    aligned_q = st["qlen"] - st["qclip"]
    assert (hs.as_ == aligned_q - 2 * hs.nm).all()
    sizes = np.diff(hs.group_off)
    assert sizes.min() >= 1 and sizes.max() <= 16 and 4.5 < sizes.mean() < 5.5


def test_grouping_rules():
    import msamtools_amd as m
    import samio
    from conftest import fixture_path
    hdr, rec = samio.read_sam(fixture_path("besthit.sam"))
    off = m.filter_pools(rec)
    names = [rec.name(i) for i in range(rec.n)]
    assert off[0] == 0 and off[-1] == rec.n
    for g in range(len(off) - 1):
        assert len(set(names[off[g]:off[g + 1]])) == 1
    assert len(off) - 1 == len(dict.fromkeys(names))
    hdr, rec = samio.read_sam(fixture_path("profile_unmapped.sam"))
    assert m.profile_pools(rec).tolist() == [0, 2]


def _switch_lists():
    text = open(os.path.join(ROOT, "README.md")).read()
    prod = re.search(r"<!-- switches: product -->(.*?)<!-- /switches -->", text, re.S).group(1)
    dbg = re.search(r"<!-- switches: debug -->(.*?)<!-- /switches -->", text, re.S).group(1)
    names = lambda t: set(re.findall(r"MSX_[A-Z][A-Z0-9_]+", t))
    return names(prod), names(dbg) - {"MSX_DEBUG_SWITCHES", "MSX_LIB_PATH"}


def _names_in(path):
    # (constants of the header that end up in error texts are not switches)
    not_switches = {"MSX_NCCL_FLOAT64", "MSX_NCCL_SUM", "MSX_NCCL_UINT32", "MSX_POOLS_FILTER"}
    return {m.decode() for m in re.findall(rb"MSX_[A-Z][A-Z0-9_]+", open(path, "rb").read())} - not_switches


def test_the_product_knows_no_switch_that_readme_does_not_list_and_none_of_the_debug_ones():
    """switches that make the library return wrong results (MSX_SR_GATHERS) and test hooks
    (MSX_INFLATE_REFUSE, MSX_CHASE_SLOPPY) were live in the product.  They are compiled in only with -DMSX_DEBUG_SWITCHES now
    (msamtools_amd/dbg/libmsamtools_amd.so, bin/msamtools-dev); README.md lists both sets and this test holds the binaries to it."""
    from msamtools_amd import _lib
    prod, dbg = _switch_lists()
    assert prod and dbg and not (prod & dbg)
    lib = _names_in(os.path.join(ROOT, "msamtools_amd", "libmsamtools_amd.so"))
    exe = _names_in(os.path.join(ROOT, "msamtools_amd", "bin", "msamtools"))
    assert not ((lib | exe) & dbg), sorted((lib | exe) & dbg)
    assert (lib | exe) <= prod, sorted((lib | exe) - prod)
    dlib = os.path.join(ROOT, "msamtools_amd", "dbg", "libmsamtools_amd.so")
    ddev = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
    assert dbg <= (_names_in(dlib) | _names_in(ddev)), sorted(dbg - (_names_in(dlib) | _names_in(ddev)))
    # the debug library is the same ABI
    out = subprocess.check_output(["nm", "-D", "--defined-only", dlib]).decode()
    assert set(declared_functions()) <= set(re.findall(r" T (msx_[a-z0-9_]+)", out))
