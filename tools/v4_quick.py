#!/usr/bin/env python3
"""V4 at batch 8192, device-resident: prints us/step; run under rocprofv3 --kernel-trace --stats for the split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.bench_configs import v4_alone  # noqa: E402

if __name__ == "__main__":
    print(v4_alone(), flush=True)
