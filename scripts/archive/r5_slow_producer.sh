#!/bin/bash
# a fatal record while the reader waits on a silent producer: when does the command itself exit?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/slow.sh <<'EOS'
#!/bin/bash
printf '@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:r1\tLN:1000\n'
for i in $(seq 1 200000); do printf 'q%07d\t0\tr1\t10\t255\t10M\t*\t0\t0\t*\t*\tAS:i:5\n' $i; done
echo "producer silent at $(date +%s.%N)" >&2
sleep 12
EOS
bash /tmp/slow.sh | ( MSX_TRACE=1 MSX_SAM_CHUNK=1000000 MSX_BATCH_RECORDS=50000 timeout 60 msamtools_amd/bin/msamtools filter -S -p 95 - 2>&1 > /dev/null | while IFS= read -r l; do echo "$(date +%s.%N | cut -c7-16) $l"; done > /tmp/err.txt; echo "consumer exited at $(date +%s.%N | cut -c7-16)" >&2 )
cut -c1-150 /tmp/err.txt | tail -15
