#!/bin/bash
code="import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests')); sys.path.insert(0, os.path.join(os.getcwd(), 'tests', 'golden')); import numpy as np, test_gpu_configs as t
try:
    w, H = t._run_band(int(os.environ.get('II','8')), 40, 6, False); print('ok')
except Exception as e:
    print('FAIL', str(e)[:60])"
for cfg in "MOPT_TRI_EARLY=2" "MOPT_TRI_EARLY=2 II=16" "A=1"; do
  r=""; for i in 1 2 3 4 5 6 7 8 9 10; do r="$r $(env $cfg python -c "$code" 2>&1 | tail -1 | cut -c1-4)"; done; echo "$cfg: $r"
done
