#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --steps 10 --warmup 2 --no-e2e --no-coverage 2>/tmp/bench.err | tail -1 > /tmp/b.json; tail -3 /tmp/bench.err | cut -c1-300
python3 -c "
import json
d=json.loads(open('/tmp/b.json').read())
r=d['roofline']; print(d['value'], d['ms_per_step'], d.get('dist_one_rank_ms_per_step'), r['avg_launch_ms'], r['frac']); print(d.get('parity'))"
