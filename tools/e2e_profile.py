#!/usr/bin/env python3
"""Run a few-layer Qwen2.5-7B-shape decode (fused path) so that `rocprofv3 --kernel-trace --stats -- python tools/e2e_profile.py`
shows where a decode step's time goes.  usage: e2e_profile.py [layers] [attention]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
att = sys.argv[2] if len(sys.argv) > 2 else "current"
print(json.dumps(e2e.bench_decode("qwen2.5-7b", batch=4, prefill=1024, steps=16, layers=layers, fused=True, attention=att)))
