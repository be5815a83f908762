#!/bin/bash
# lag_ab.sh "ENV=VAL .." ...: headline bench line (pairs/s, ms/step, allocated / reserved GB) per environment setting
R=${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  env $cfg python3 $R/bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 --roofline-steps 0 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$cfg', d['value'], d['ms_per_step'], d['ms_per_step_median'], c['peak_hbm_gb'], c['peak_hbm_reserved_gb'], c['alloc_retries'])"
done
