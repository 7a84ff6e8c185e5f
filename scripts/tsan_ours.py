#!/usr/bin/env python3
"""ThreadSanitizer logs of the command line (scripts/archive/r5_tsan_cli.sh): the HIP / HSA runtimes are not instrumented, so
most reports lie wholly inside them (operator new / delete, their own mutexes).  Prints the reports in which one of the two
racing accesses has ITS TOP FRAME in the host code (msamtools_amd/csrc/host), and the count of the rest.
usage: tsan_ours.py LOG..."""
import collections
import re
import sys

ours, rest = collections.Counter(), 0
for f in sys.argv[1:]:
    for rep in open(f, errors="replace").read().split("==================\n"):
        if "WARNING: ThreadSanitizer" not in rep:
            continue
        kind = re.search(r"WARNING: ThreadSanitizer: ([^(]+)", rep).group(1).strip()
        tops = []
        for b in rep.split("\n\n"):
            if re.match(r"\s*(Read|Write|Previous|Atomic)", b.strip()):
                m = re.search(r"#0 (\S+) (\S+)", b)
                if m:
                    tops.append((m.group(1), m.group(2)))
        mine = [t for t in tops if "msamtools_amd/csrc/host" in t[1]]
        if mine:
            ours[(kind,) + tuple(f"{a} {b.split('/')[-1]}" for a, b in mine)] += 1
        else:
            rest += 1
for k, v in ours.most_common():
    print(v, *k)
print(f"{sum(ours.values())} report(s) with an access in the host code; {rest} wholly inside uninstrumented libraries")
