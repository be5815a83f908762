#!/usr/bin/env python3
"""World fixtures for tests/test_world_gpu.py: the CPU ORACLE (oracle/leccr_oracle.py, pinned to the reference by the other fixtures)
run under W-rank gloo - W real processes, AllGather with the slice-only backward (models/xvlm.py:50-70), DDP-style SUM / W of the
parameter gradients (image_Retrieval_caption.py:390-392) - on the seeded tiny-tower model and global batch of that test file.

    python tests/golden/make_world_golden.py        # writes tests/golden/world_oracle_w{4,8}_g{1024,2048}.npz  (CPU, ~1 min)

Kept per fixture: losses [W, 5]; per gradient tensor its norm ('n/<name>') and a strided sample of <= 1024 elements ('s/<name>').
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from tests.test_world_gpu import oracle_world, pack_world_fixture, world_fixture_name
    out = os.path.dirname(os.path.abspath(__file__))
    for world, G in ((4, 1024), (8, 1024), (8, 2048)):
        losses, grads = oracle_world(world, G)
        path = os.path.join(out, world_fixture_name(world, G))
        np.savez_compressed(path, **pack_world_fixture(losses, grads))
        print(path, os.path.getsize(path), losses[0])


if __name__ == '__main__':
    main()
