#!/usr/bin/env python3
"""Launch ONE roofline entry of pdgn_amd.roofline a few times (the target of per-entry rocprofv3 --pmc passes).
usage: roofline_entry.py <entry function name>   (see pdgn_amd.roofline.ENTRIES)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import roofline
roofline._time_us.__defaults__ = (5, 2)          # 2 warm-up + 5 timed launches
fn = getattr(roofline, sys.argv[1])
print(json.dumps(fn(35, 128, torch.device("cuda", 0))))
