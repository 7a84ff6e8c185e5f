#!/bin/bash
# round 6: exec to main -- what the dynamic loader costs a command (no HIP call made)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
for rep in 1 2 3 4 5; do
  a=$EPOCHREALTIME; $B help > /dev/null 2>&1; b=$EPOCHREALTIME
  python3 -c "print('help: %.1f ms' % (($b-$a)*1e3))"
done
LD_DEBUG=statistics $B help 2>&1 | grep -i "total startup\|relocation\|load" | head -8
ldd $B | head -30
