#!/usr/bin/env python3
"""One-off refactoring tool (round 4): cut msamtools_amd/csrc/host/msh_cli.c into one file per command + the pipeline,
moving what several of them use into msh_cli.h.  Kept for the record of how the split was made; not part of any build."""
import re
import sys

SRC = "msamtools_amd/csrc/host/msh_cli.c"
text = open(SRC).read()
lines = text.split("\n")

# ---- top-level items: (first line, last line) by a brace-depth scan that skips strings, chars and comments ----
items = []
depth = 0
state = None          # None | "str" | "chr" | "blk" (block comment)
start = None
i = 0
pos_line = 0
cur_start = 0
for ln, line in enumerate(lines):
    j = 0
    n = len(line)
    line_comment = False
    while j < n:
        c = line[j]
        if state == "blk":
            if line.startswith("*/", j):
                state = None
                j += 2
                continue
        elif state == "str":
            if c == "\\":
                j += 2
                continue
            if c == '"':
                state = None
        elif state == "chr":
            if c == "\\":
                j += 2
                continue
            if c == "'":
                state = None
        else:
            if line.startswith("//", j):
                break
            if line.startswith("/*", j):
                state = "blk"
                j += 2
                continue
            if c == '"':
                state = "str"
            elif c == "'":
                state = "chr"
            elif c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
        j += 1
    # an item ends on a line where depth is 0, we are outside comments, and the line ends a statement / definition / directive
    if depth == 0 and state is None:
        s = line.rstrip()
        if s == "" or s.endswith(";") or s.endswith("}") or s.endswith("*/") or s.startswith("#") and not s.endswith("\\"):
            items.append((cur_start, ln))
            cur_start = ln + 1
if cur_start < len(lines):
    items.append((cur_start, len(lines) - 1))

# merge: a comment block or blank lines in front of a definition belong to it
merged = []
pending = None
for a, b in items:
    body = "\n".join(lines[a:b + 1]).strip()
    only_comment = body == "" or (body.startswith("/*") and body.endswith("*/") and body.count("/*") == 1) or all(
        l.strip() == "" or l.strip().startswith("//") for l in lines[a:b + 1])
    if pending is None:
        pending = [a, b]
    else:
        pending[1] = b
    if not only_comment:
        merged.append(tuple(pending))
        pending = None
if pending:
    merged.append(tuple(pending))


def item_text(it):
    return "\n".join(lines[it[0]:it[1] + 1])


def classify(it):
    """(kind, name): func / typedef / define / var / other"""
    t = item_text(it)
    code = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    code = "\n".join(l for l in code.split("\n") if not l.strip().startswith("//")).strip()
    if code.startswith("#define"):
        return "define", re.match(r"#define\s+(\w+)", code).group(1)
    if code.startswith("#include") or code.startswith("#"):
        return "directive", None
    if code.startswith("typedef"):
        m = re.search(r"(\w+)\s*;\s*$", code)
        return "typedef", m.group(1) if m else None
    if code.startswith("struct") and code.rstrip().endswith("};"):
        m = re.match(r"struct\s+(\w+)", code)
        return "struct", m.group(1)
    m = re.match(r"(?:static\s+)?(?:__thread\s+)?(?:inline\s+)?[\w\s\*]+?\b(\w+)\s*\([^;{]*\)\s*\{", code, flags=re.S)
    if m and "{" in code and code.rstrip().endswith("}"):
        return "func", m.group(1)
    if code.endswith(";"):
        m = re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*(?:=[^;]*)?;\s*$", code)
        return "var", m.group(1) if m else None
    return "other", None


# ---- which file every item goes to: by the line ranges of the old file ----
def find_line(pat, after=0):
    for k in range(after, len(lines)):
        if re.search(pat, lines[k]):
            return k
    raise SystemExit("pattern not found: " + pat)


L_rbatch = find_line(r"^/\* record batch: BAM blobs")
L_filter_help = find_line(r"^static void filter_help")
L_reader = find_line(r"^typedef struct \{", L_filter_help)       # reader
L_bulk = find_line(r"^/\* ---- bulk path for BAM input")
L_rescore = find_line(r"^/\* --rescore: drop the first AS")
L_profile = find_line(r"^static void profile_help") - 3
L_filter_pipe = find_line(r"^/\* ---- filter over the pipeline")
L_profile_pipe = find_line(r"^/\* ---- profile over the pipeline")
L_cov = find_line(r"^static void coverage_help") - 3
L_usage = find_line(r"^static int usage") - 3
L_dev = find_line(r"^/\* Host I/O self-test")
L_main = find_line(r"^int main\(")


def target(line_no):
    if line_no < L_filter_help - 3:
        return "msh_common.c"
    if line_no < L_reader - 0:
        return "msh_filter.c"
    if line_no < L_rescore:
        return "msh_pipeline.c"
    if line_no < L_profile:
        return "msh_filter.c"
    if line_no < L_filter_pipe:
        return "msh_profile.c"
    if line_no < L_profile_pipe:
        return "msh_filter.c"
    if line_no < L_cov:
        return "msh_profile.c"
    if line_no < L_usage:
        return "msh_coverage.c"
    if line_no < L_dev:
        return "msh_main.c"
    if line_no < L_main:
        return "msh_dev.c"
    return "MAIN"


files = {}
info = []
for it in merged:
    kind, name = classify(it)
    tgt = target(it[1] if kind != "other" else it[0])
    # the item's definition line decides (comments in front of it travel along)
    info.append({"it": it, "kind": kind, "name": name, "file": tgt, "text": item_text(it)})

# the head of the old file (comment + includes) is regenerated
head_end = find_line(r"^#define QNAME_GROUP_CHECK_RECORDS")
info = [x for x in info if x["it"][0] >= head_end]

by_file = {}
for x in info:
    by_file.setdefault(x["file"], []).append(x)

# ---- what is used outside its own file ----
def uses(name, fname):
    pat = re.compile(r"\b" + re.escape(name) + r"\b")
    for f, xs in by_file.items():
        if f == fname:
            continue
        for x in xs:
            if x["name"] == name and x["kind"] in ("func", "typedef", "define", "var", "struct"):
                continue
            if pat.search(x["text"]):
                return True
    return False


header_items = []
protos = []
externs = []
for x in info:
    if not x["name"]:
        continue
    if x["kind"] in ("typedef", "define", "struct"):
        # types and macros: into the header when another file uses them (or a type the header needs: second pass below)
        if uses(x["name"], x["file"]):
            x["to_header"] = True
    elif x["kind"] == "func":
        if uses(x["name"], x["file"]) and x["name"] != "main":
            x["shared"] = True
    elif x["kind"] == "var":
        if uses(x["name"], x["file"]):
            x["shared"] = True

# a header item may need other types / macros defined before it: pull those in too (transitively)
changed = True
while changed:
    changed = False
    hdr_text = "\n".join(x["text"] for x in info if x.get("to_header"))
    shared_sigs = "\n".join(x["text"].split("{")[0] for x in info if x.get("shared") and x["kind"] == "func")
    for x in info:
        if x["kind"] in ("typedef", "define", "struct") and not x.get("to_header") and x["name"]:
            if re.search(r"\b" + re.escape(x["name"]) + r"\b", hdr_text + "\n" + shared_sigs):
                x["to_header"] = True
                changed = True

out = {}
for x in info:
    if x.get("to_header"):
        header_items.append(x)
        continue
    t = x["text"]
    if x.get("shared"):
        if x["kind"] == "func":
            sig = t[t.index(re.search(r"^(static\s+)?[\w\s\*]*\b" + re.escape(x["name"]) + r"\s*\(", t, flags=re.M).group(0)):]
            sig = sig[:sig.index("{")].strip()
            sig = re.sub(r"^static\s+(inline\s+)?", "", sig)
            protos.append((x["file"], sig + ";"))
            t = re.sub(r"^static\s+(inline\s+)?(?=[\w\s\*]*\b" + re.escape(x["name"]) + r"\s*\()", "", t, count=1, flags=re.M)
        else:
            decl = re.sub(r"/\*.*?\*/", "", t, flags=re.S).strip()
            decl_line = [l for l in decl.split("\n") if l.strip()][-1].strip() if decl else ""
            m = re.match(r"static\s+(.*?)\s*(=.*)?;$", decl_line)
            if m:
                # (several variables in one declaration stay together)
                externs.append((x["file"], "extern " + m.group(1) + ";"))
                t = t.replace(decl_line, decl_line.replace("static ", "", 1), 1)
    out.setdefault(x["file"], []).append(t)

HEAD = {
    "msh_common.c": "msh_common.c -- what the commands share: the calling thread's context, stage timers, the device list, record batches\n * (BAM blobs + the SoA view the kernels read) and the QNAME-grouping preflight (msam_helper.c:78-137, 295-484).",
    "msh_pipeline.c": "msh_pipeline.c -- the decode | device | encode pipeline under the commands: batch slots, the decode stage (BGZF blocks\n * inflated here or handed to the device compressed, the speculative record chase, SAM text), page-locking, queues.\n * Counterpart of the read loop msam_helper.c:246-268 feeding msam_filter.c:119-186 / msam_profile.c:222-234.",
    "msh_filter.c": "msh_filter.c -- `msamtools filter` (msam_filter.c:304-497): options and their validation messages, the device stage and\n * the writer stage over the pipeline, --rescore, --profile-out.",
    "msh_profile.c": "msh_profile.c -- `msamtools profile` (msam_profile.c:554-990): options, --genome features, the device stage over the\n * pipeline, post-processing and the text report (mMatrix.c:359-376), shared with `filter --profile-out`.",
    "msh_coverage.c": "msh_coverage.c -- `msamtools coverage` (msam_coverage.c:225-380) over the pipeline.",
    "msh_main.c": "msh_main.c -- the program's entry: command dispatch as msamtools.c:8-49 (filter, profile, coverage, help).",
    "msh_dev.c": "msh_dev.c -- `msamtools-dev`: developer and test commands that are NOT part of the reference's surface and not in the\n * product binary: synth (deterministic BAM generator), recode, digest, pipetest, restream, rawtest, keyorder.",
}

hdr = ["/* msh_cli.h -- internals shared by the files of the command line (msh_common.c, msh_pipeline.c, msh_filter.c,",
       " * msh_profile.c, msh_coverage.c, msh_main.c, msh_dev.c). */", "#ifndef MSH_CLI_H", "#define MSH_CLI_H",
       '#include "msh.h"', "", "#include <errno.h>", "#include <fcntl.h>", "#include <getopt.h>", "#include <pthread.h>",
       "#include <sys/mman.h>", "#include <math.h>", "#include <time.h>", "#include <unistd.h>", "#include <zlib.h>", ""]
for x in header_items:
    hdr.append(x["text"])
hdr.append("")
last = None
for f, e in externs:
    if f != last:
        hdr.append(f"/* {f} */")
        last = f
    hdr.append(e)
last = None
for f, p in protos:
    if f != last:
        hdr.append(f"/* {f} */")
        last = f
    hdr.append(p)
hdr.append("#endif")
D = "msamtools_amd/csrc/host/"
open(D + "msh_cli.h", "w").write("\n".join(hdr) + "\n")
main_items = out.pop("MAIN", [])
for f, parts in out.items():
    body = "\n".join(parts)
    open(D + f, "w").write("/*\n * " + HEAD[f] + "\n */\n#include \"msh_cli.h\"\n\n" + body.strip("\n") + "\n")
open("/tmp/old_main.c", "w").write("\n".join(main_items))
print({f: len(p) for f, p in out.items()}, "header items", len(header_items), "protos", len(protos), "externs", len(externs))
