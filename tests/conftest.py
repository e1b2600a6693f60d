import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')
    # The CPU side of the suite (kernel emulation, oracle) is thousands of small torch ops: on the GPU boxes' 128-256 hardware threads the
    # default intra-op pool makes them slower, not faster (bench.py's cpu_baseline sweep: 16 threads is the fastest setting there).
    # MV2_TEST_THREADS overrides (0 = torch's default).
    import torch
    n = int(os.environ.get('MV2_TEST_THREADS', '16'))
    if n > 0:
        torch.set_num_threads(max(1, min(n, os.cpu_count() or n)))
