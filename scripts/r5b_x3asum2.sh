#!/bin/bash
O=gpurun_out/x3asum; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "sums_the_bias_gradient_from_the_split_copy" > $O/tests_kernel2.log 2>&1; echo "kernel tests rc $?"; tail -3 $O/tests_kernel2.log; grep "^E  " $O/tests_kernel2.log | head
