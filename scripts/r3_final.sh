#!/bin/bash
# r3_final.sh : the round's closing measurements on one box (bench lines, kernel-stats profiles, PMC passes of the dW kernel)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
python3 $R/bench.py > $O/r3f_bench_b16.json 2> $O/r3f_bench_b16.err; tail -c 600 $O/r3f_bench_b16.json; echo
python3 $R/bench.py --clip ViT-B/32 --no-cpu-baseline > $O/r3f_bench_b32.json 2> $O/r3f_bench_b32.err
python3 $R/bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 1 > $O/r3f_bench_E_bf16.json 2> $O/r3f_bench_E_bf16.err
python3 $R/bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 1 > $O/r3f_bench_E_fp8.json 2> $O/r3f_bench_E_fp8.err
for f in b16 b32 E_bf16 E_fp8; do python3 -c "import sys,json; d=json.loads(open('$O/r3f_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['achieved'], r['peak'], r['frac'], d['config'].get('peak_hbm_gb'), d['config'].get('peak_hbm_reserved_gb'))"; done
python3 $R/scripts/gemm_fp8_dw_bench.py > $O/r3f_gemm_fp8_dw.txt 2>&1
bash $R/scripts/profile_bench.sh r3f
bash $R/scripts/profile_bench.sh r3f_ss LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0
bash $R/scripts/profile_cfgE.sh fp8
SHAPES=dw bash $R/scripts/pmc_gemm_shapes.sh > /dev/null 2>&1
cp $O/pmc_gemm_shapes.txt $O/r3f_pmc_dw.txt
