#!/usr/bin/env python3
"""Pack the sample data sets that pyMRA ships (pyMRA/data/{small,large}/{locs,y,y_obs}.npy: 10x10 and 100x100
exponential-GRF samples, the inputs of README example 1 and of KAT4) into pymra_amd/data/<size>.npz so that
pymra_amd.DataLoader.load_data has something to load where the reference is not installed.  Data only.
Run in the build container:  python tools/pack_sample_data.py [/root/reference]"""
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for size in ("small", "large"):
    d = os.path.join(ref, "pyMRA", "data", size)
    arrs = {k: np.load(os.path.join(d, k + ".npy")) for k in ("locs", "y", "y_obs")}
    out = os.path.join(here, "pymra_amd", "data", size + ".npz")
    np.savez_compressed(out, **arrs)
    print(size, {k: v.shape for k, v in arrs.items()}, "NaN in y_obs:", int(np.isnan(arrs["y_obs"]).sum()), "->", out)
