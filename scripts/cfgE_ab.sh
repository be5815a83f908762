#!/bin/bash
# cfgE_ab.sh "ENV=VAL ..." ... : config-E-geometry bench line (value, ms/step) per environment setting (A/B of switches), fp8 unless DT is set
R=${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  env $cfg python3 $R/bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype ${DT:-fp8} --steps 4 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0 2>>$R/gpurun_out/cfgE_ab.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$cfg', d['dtype'], d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'alloc', c.get('peak_hbm_gb'), 'reserved', c.get('peak_hbm_reserved_gb'), 'GB retries', c.get('alloc_retries'))"
done
