"""Sample data sets, with the call surface of pyMRA/DataLoader.py:5-19.

``load_data(size, include_truth)``: ``size`` is "small" (10 x 10 grid) or "large" (100 x 100); returns
``(y, locs, y_obs)`` when ``include_truth`` is true.  The reference's ``include_truth=False`` branch returns
``(y, locs)`` with ``y`` never assigned (an ``UnboundLocalError``, DataLoader.py:19); the evident intent -
observations and locations without the truth - is what is returned here: ``(y_obs, locs)``.
The arrays are the reference's own sample files, repacked by tools/pack_sample_data.py.
"""
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_data(size="small", include_truth=False):
    if size not in ["small", "large"]:
        raise ValueError("size has to be 'small' or 'large'")
    with np.load(os.path.join(_DIR, size + ".npz")) as f:
        y_obs, locs, y = f["y_obs"], f["locs"], f["y"]
    if include_truth:
        return (y, locs, y_obs)
    return (y_obs, locs)
