#!/usr/bin/env python3
"""One-off converter: modelnet40_ply_hdf5_2048/ply_data_{train,test}*.h5 -> ply_data_*.npz with the same datasets
(`data` float32 [n,2048,3], `normal` float32 [n,2048,3], `label` uint8 [n,1]), next to the h5 files.

h5py is NOT in the MI355X image (and learning3d_amd never imports it): run this once wherever h5py exists, then
learning3d_amd.data_utils.disk_feed reads the .npz files (reference reader: data_utils/dataloaders.py:31-48).

    python tools/modelnet40_h5_to_npz.py /path/to/data/modelnet40_ply_hdf5_2048
"""
import glob
import os
import sys

import numpy as np


def main(root):
    import h5py
    files = sorted(glob.glob(os.path.join(root, "ply_data_*.h5")))
    if not files:
        raise SystemExit(f"no ply_data_*.h5 under {root}")
    for fn in files:
        with h5py.File(fn, "r") as f:
            arrays = {k: f[k][:] for k in ("data", "normal", "label") if k in f}
        out = fn[:-3] + ".npz"
        np.savez(out, **arrays)
        print(out, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
