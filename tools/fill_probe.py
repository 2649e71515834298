#!/usr/bin/env python3
"""The fp32 fill (sq_bpmatrix_fill) of 256 x S1000 a few times: the leg of bench.py's fill roofline, alone (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.fill_leg()
print("fill: %.1f GB/s (%.4f of HBM peak), %.4f ms per launch" % (r["achieved"], r["frac"], r["ms_per_fill"]))
