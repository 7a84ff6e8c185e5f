cd $GRAFT_REPO_ROOT
for k in "random_cuts" "coverage_fuzz or random_cuts" "long_runs or random_cuts" "whole_sample_coverage_parity or random_cuts" "synth_coverage or golden or random_cuts" "collected"; do
  timeout 300 python -m pytest -m gpu -x -q tests/test_gpu_parity.py -k "$k" > /tmp/o.log 2>&1; rc=$?
  echo "[-k $k] rc=$rc $(grep -E 'passed|failed' /tmp/o.log | tail -1) $(head -1 /tmp/o.log | cut -c1-60)"
done
