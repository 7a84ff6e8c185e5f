#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv -> per-kernel averages of every counter (one JSON object on stdout)."""
import collections
import csv
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, sys.stdout, indent=1)
