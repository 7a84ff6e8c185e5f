"""Independent SAM-text / BAM readers for the test-suite (pure Python + numpy).

They turn a file into the structure-of-arrays record stream that both the CPU
oracle (oracle/msx_oracle.h: orc_records) and the product's C-ABI
(include/msamtools_amd.h: msx_batch) consume.  Written from the SAM/BAM
specification (SAMv1 sections 1.4, 4.2), independently of the product's C
reader in msamtools_amd/csrc/host, so that the two can be cross-checked.
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass, field

import numpy as np

HAS_MD, HAS_NM, HAS_AS = 1, 2, 4
CIGAR_OPS = "MIDNSHP=X"


@dataclass
class Header:
    text: str = ""
    target_name: list = field(default_factory=list)
    target_len: list = field(default_factory=list)

    @property
    def n_targets(self):
        return len(self.target_name)

    def sort_order(self):
        for line in self.text.split("\n"):
            if line.startswith("@HD"):
                for f in line.split("\t")[1:]:
                    if f.startswith("SO:"):
                        return f[3:]
        return None


@dataclass
class Records:
    """SoA record stream; arrays are numpy, C-contiguous."""
    qname_off: np.ndarray
    qname: np.ndarray
    flag: np.ndarray
    rflags: np.ndarray
    tid: np.ndarray
    pos: np.ndarray
    cigar_off: np.ndarray
    cigar: np.ndarray
    md_off: np.ndarray
    md: np.ndarray
    nm: np.ndarray
    as_: np.ndarray

    @property
    def n(self):
        return int(self.flag.shape[0])

    def name(self, i):
        return bytes(self.qname[self.qname_off[i]:self.qname_off[i + 1]]).decode()

    def md_str(self, i):
        return bytes(self.md[self.md_off[i]:self.md_off[i + 1]]).decode()

    def cigar_str(self, i):
        ops = self.cigar[self.cigar_off[i]:self.cigar_off[i + 1]]
        return "".join(f"{int(c) >> 4}{CIGAR_OPS[int(c) & 15]}" for c in ops) or "*"


class _Builder:
    def __init__(self):
        self.qn, self.flag, self.rfl, self.tid, self.pos = [], [], [], [], []
        self.cig, self.md, self.nm, self.as_ = [], [], [], []

    def add(self, qname, flag, tid, pos, cigar, md, nm, as_):
        rf = 0
        if md is not None:
            rf |= HAS_MD
        if nm is not None:
            rf |= HAS_NM
        if as_ is not None:
            rf |= HAS_AS
        self.qn.append(qname)
        self.flag.append(flag)
        self.rfl.append(rf)
        self.tid.append(tid)
        self.pos.append(pos)
        self.cig.append(cigar)
        self.md.append(md or b"")
        self.nm.append(_i32(nm or 0))
        self.as_.append(_i32(as_ or 0))

    def build(self):
        def csr(items, dtype):
            off = np.zeros(len(items) + 1, dtype=np.uint32)
            if items:
                off[1:] = np.cumsum([len(x) for x in items])
            flat = np.array([v for x in items for v in x], dtype=dtype) if off[-1] else np.zeros(0, dtype=dtype)
            return off, flat
        qoff, q = csr(self.qn, np.uint8)
        coff, c = csr(self.cig, np.uint32)
        moff, m = csr(self.md, np.uint8)
        return Records(
            qname_off=qoff, qname=q,
            flag=np.array(self.flag, dtype=np.uint16),
            rflags=np.array(self.rfl, dtype=np.uint8),
            tid=np.array(self.tid, dtype=np.int32),
            pos=np.array(self.pos, dtype=np.int32),
            cigar_off=coff, cigar=c, md_off=moff, md=m,
            nm=np.array(self.nm, dtype=np.int32),
            as_=np.array(self.as_, dtype=np.int32),
        )


def _i32(v):
    """(int32_t) bam_aux2i(): truncate an int64 value to int32 like the reference does."""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def parse_cigar_text(s):
    if s == "*":
        return []
    out, num = [], 0
    for ch in s:
        if ch.isdigit():
            num = num * 10 + ord(ch) - 48
        else:
            out.append((num << 4) | CIGAR_OPS.index(ch))
            num = 0
    return out


def read_sam(path):
    hdr = Header()
    b = _Builder()
    lines = open(path, "r").read().split("\n")
    name2tid = {}
    htext = []
    for line in lines:
        if not line:
            continue
        if line.startswith("@"):
            htext.append(line)
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.split("\t")[1:])
                name2tid[f["SN"]] = len(hdr.target_name)
                hdr.target_name.append(f["SN"])
                hdr.target_len.append(int(f["LN"]))
            continue
        f = line.split("\t")
        tid = -1 if f[2] == "*" else name2tid[f[2]]
        md = nm = as_ = None
        for tag in f[11:]:
            t, ty, val = tag.split(":", 2)
            # bam_aux_get: the FIRST occurrence wins
            if t == "MD" and md is None:
                md = val.encode()
            elif t == "NM" and nm is None:
                nm = int(val) if ty == "i" else 0
            elif t == "AS" and as_ is None:
                as_ = int(val) if ty == "i" else 0
        b.add(f[0].encode(), int(f[1]), tid, int(f[3]) - 1, parse_cigar_text(f[5]), md, nm, as_)
    hdr.text = "\n".join(htext) + ("\n" if htext else "")
    return hdr, b.build()


_AUX_FIXED = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
_AUX_INT = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}


def _walk_aux(buf):
    """Yield (tag, type, value) for a BAM aux block; value is int for integer
    types, bytes for Z, None otherwise (bam_aux2i gives 0 for those)."""
    p, n = 0, len(buf)
    while p + 3 <= n:
        tag = buf[p:p + 2]
        ty = chr(buf[p + 2])
        p += 3
        if ty in _AUX_INT:
            sz = _AUX_FIXED[ty]
            yield tag, ty, struct.unpack_from(_AUX_INT[ty], buf, p)[0]
            p += sz
        elif ty in _AUX_FIXED:
            yield tag, ty, None
            p += _AUX_FIXED[ty]
        elif ty in "ZH":
            e = buf.index(b"\0", p)
            yield tag, ty, bytes(buf[p:e])
            p = e + 1
        elif ty == "B":
            sub = chr(buf[p])
            cnt = struct.unpack_from("<I", buf, p + 1)[0]
            yield tag, ty, None
            p += 5 + cnt * _AUX_FIXED[sub]
        else:
            raise ValueError(f"bad aux type {ty!r}")


def read_bam(path):
    raw = gzip.open(path, "rb").read()   # BGZF = concatenated gzip members
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    hdr = Header(text=raw[8:8 + l_text].split(b"\0")[0].decode())
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        hdr.target_name.append(raw[p + 4:p + 4 + l_name - 1].decode())
        hdr.target_len.append(struct.unpack_from("<i", raw, p + 4 + l_name)[0])
        p += 8 + l_name
    b = _Builder()
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        rec = raw[p + 4:p + 4 + bs]
        p += 4 + bs
        tid, pos, l_rn, _mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        q = 32
        qname = rec[q:q + l_rn - 1]
        q += l_rn
        cigar = list(struct.unpack_from(f"<{n_cig}I", rec, q))
        q += 4 * n_cig + (l_seq + 1) // 2 + l_seq
        md = nm = as_ = None
        # a CIGAR of more than 65535 operations: placeholder <l_seq>S<ref_len>N + the real one in the first CG tag if that is
        # B,I / B,i with at least n_cigar elements (htslib: bam_tag2cigar under sam_read1, msam_helper.c:246-268)
        if n_cig >= 1 and tid >= 0 and pos >= 0 and (cigar[0] & 15) == 4 and (cigar[0] >> 4) == l_seq:
            a, n_aux = q, len(rec)
            while a + 3 <= n_aux:
                ty = chr(rec[a + 2])
                if rec[a:a + 2] == b"CG":
                    if ty == "B" and chr(rec[a + 3]) in "Ii":
                        cnt = struct.unpack_from("<I", rec, a + 4)[0]
                        if n_cig <= cnt < (1 << 29):
                            cigar = list(struct.unpack_from(f"<{cnt}I", rec, a + 8))
                    break
                if ty in _AUX_FIXED:
                    a += 3 + _AUX_FIXED[ty]
                elif ty in "ZH":
                    a = rec.index(b"\0", a + 3) + 1
                else:
                    a += 8 + struct.unpack_from("<I", rec, a + 4)[0] * _AUX_FIXED[chr(rec[a + 3])]
        for tag, ty, val in _walk_aux(rec[q:]):
            if tag == b"MD" and md is None:
                md = val if ty == "Z" else b""
            elif tag == b"NM" and nm is None:
                nm = val if isinstance(val, int) else 0
            elif tag == b"AS" and as_ is None:
                as_ = val if isinstance(val, int) else 0
        b.add(qname, flag, tid, pos, cigar, md, nm, as_)
    return hdr, b.build()


def read_any(path):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    return read_bam(path) if magic == b"\x1f\x8b" else read_sam(path)
