#!/bin/bash
# profile_cfgE.sh DTYPE : rocprofv3 kernel stats of the config-E-geometry step (ViT-L/14 + XLM-R-large, batch 256) -> gpurun_out/prof_E_DTYPE/
dt=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_E_$dt -o E_$dt -- python3 $R/bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0 > $R/gpurun_out/prof_E_$dt.log 2>&1
tail -1 $R/gpurun_out/prof_E_$dt.log | cut -c1-160
