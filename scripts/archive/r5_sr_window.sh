STEP="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-coverage --no-dist-leg"
for w in 1 0; do
  echo "== MSX_SR_WINDOW=$w"
  MSX_SR_WINDOW=$w MSX_SR_CLASSES=1 python3 $STEP 2> /tmp/err.$w | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'), d.get('parity',{}).get('ok'), d.get('parity',{}).get('max_rel_err'))
for k,v in d['roofline'].get('per_kernel',{}).items():
    if k in ('k_share_reduce','k_list_order','k_prop_apply'): print('  ',k,v.get('ms_per_step'),v.get('launches'))
"
  grep "derived store" /tmp/err.$w | tail -2
done
