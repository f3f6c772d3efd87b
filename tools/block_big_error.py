#!/usr/bin/env python3
"""The F = 128, N = 512 bilateral block (tests/golden/deconv_bilateral_big.npz, from the imported reference, fp32 and fp64 runs) in the
three arithmetic modes of the contractions: per quantity  error vs the reference's fp32 run | vs its fp64 run | the fp32 reference's
own distance from its fp64 run  (max |a - b| / max |b|), and which contraction shapes ran (GPU box: python3 tools/block_big_error.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_deconv as T  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "deconv_bilateral_big.npz"))
res = {}
for mode in ("x2", "x3", "fp32"):
    res[mode], log = T.block_big_errors(g, mode)
    if mode == "x2":
        print("contractions:", sorted(set(log)))
names = [n for n in res["x2"] if isinstance(res["x2"][n], tuple)]
print("%-34s %-32s %-32s %-32s %s" % ("quantity", "x2: vs ref32 | vs ref64", "x3", "fp32", "ref32 vs ref64"))
for n in names:
    print("%-34s %s %.2e" % (n, " ".join("%.2e | %.2e         " % res[m][n][:2] for m in ("x2", "x3", "fp32")), res["x2"][n][2]))
for m in ("x2", "x3", "fp32"):
    print(m, "y elementwise (in units of 1e-4 rel + 2e-5): %.3f" % res[m]["y_elementwise"],
          " worst norm: %.2e  worst zero-bias residue: %.2e  worst buffer (units of 1e-4 rel + 1e-5): %.3f" %
          (max(v for k, v in res[m].items() if k.startswith("norm.")), max(v for k, v in res[m].items() if k.startswith("zero.")),
           max(v for k, v in res[m].items() if k.startswith("stat."))))
