#!/bin/bash
# round 5: the whole GPU suite in one process, progress to a file (a line per test) so that the run is never silent
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 ${SUITE_TIMEOUT:-1150} python -m pytest tests/ -q -m gpu -p no:cacheprovider ${SUITE_ARGS:-} -v > gpurun_out/r5_fullsuite.log 2>&1
rc=$?
tail -15 gpurun_out/r5_fullsuite.log
cat gpurun_out/graded_eval_*.txt 2>/dev/null | cut -c1-400
exit $rc
