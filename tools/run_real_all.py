#!/usr/bin/env python
"""Launcher of the dataset runner (icra20-hand-object-pose_amd/run_real_all.py).

    python tools/run_real_all.py --root DIR [--synthetic 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_real_all.py --root DIR
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hop_loader  # noqa: E402

hop_loader.load()
from hop_amd import run_real_all  # noqa: E402

if __name__ == "__main__":
    run_real_all.main()
