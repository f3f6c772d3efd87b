#!/usr/bin/env python3
"""Launch only the roofline kernels (pdgn_amd.roofline.measure) -- the target of the rocprofv3 --pmc passes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import roofline
print(json.dumps(roofline.measure(35, 128, torch.device("cuda", 0))))
