for e in 1 2 4 5 10; do echo "== exp $e"; LECCR_LIB=gpurun_variants/libleccr_w4e$e.so VARIANT=3 ONLY="fc1,fc2,sq 4096,sq 8192" timeout -k 10 100 python scripts/gemm_bench.py 2>&1 | grep TFLOP; done
