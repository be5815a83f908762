#!/bin/bash
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python3 -m pytest tests/test_eval_gpu.py -x -q -m gpu -k "as_yielded or synthetic_loader" 2>&1 | tail -2
