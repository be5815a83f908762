#!/bin/bash
REPS=20 python3 scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | head -10
