#!/usr/bin/env python3.9
"""Generates tests/golden/voxel_cache/*.h5 with the REFERENCE's own writer (data/utils/generic.py:49-55 `np_array_to_h5`:
h5py.create_dataset('voxel_grid', compression=32001, compression_opts=(0, 0, 0, 0, 1, 1, 5)) on the real libhdf5 + hdf5-blosc filter + c-blosc.

Run in the build container with the conda interpreter that has them (the system Python 3.10 has neither h5py nor blosc):
    /opt/conda/bin/python3.9 tests/golden/make_voxel_cache_golden.py
(h5py 3.3.0 / HDF5 1.10.6; the Blosc filter 32001 is the one PyTables 3.6.1 bundles and registers with libhdf5 on import: c-blosc 1.20.1.)
The arrays are not stored: tests/test_voxel_cache.py regenerates them from the seeds below (`golden_array`)."""
import json
import os
import sys
import types

import numpy

numpy.typeDict = numpy.sctypeDict          # PyTables 3.6.1 predates numpy 1.24
import tables                              # noqa: F401  (registers HDF5 filter 32001 = Blosc)
import h5py

sys.modules["cv2"] = types.ModuleType("cv2")          # generic.py imports cv2 for the flow PNGs only
sys.path.insert(0, "/root/reference")
from data.utils.generic import np_array_to_h5, h5_to_np_array     # the reference's functions themselves

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxel_cache")
CASES = {                                   # name: (shape, seed)
    "dsec_15bins_small": ((15, 60, 80), 11),
    "ragged_5bins": ((5, 33, 47), 12),
    "multiflow_65bins": ((65, 24, 32), 13),
}


def golden_array(shape, seed):
    """Sparse signed voxel grid: ~30 % non-zero cells, N(0, 1) values (float32) -- tests/test_voxel_cache.py carries the same function."""
    rs = numpy.random.RandomState(seed)
    return (rs.standard_normal(shape) * (rs.uniform(size=shape) < 0.3)).astype(numpy.float32)


def main():
    assert h5py.h5z.filter_avail(32001), "Blosc filter not registered"
    os.makedirs(HERE, exist_ok=True)
    meta = {"h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version, "pytables": tables.__version__,
            "blosc": str(tables.which_lib_version("blosc")[1]), "cases": {}}
    for name, (shape, seed) in CASES.items():
        a = golden_array(shape, seed)
        path = os.path.join(HERE, name + ".h5")
        if os.path.exists(path):
            os.remove(path)
        np_array_to_h5(a, __import__("pathlib").Path(path))
        back = h5_to_np_array(__import__("pathlib").Path(path))
        assert back.dtype == a.dtype and (back == a).all()
        with h5py.File(path, "r") as f:
            d = f["voxel_grid"]
            meta["cases"][name] = {"shape": list(shape), "seed": seed, "chunks": list(d.chunks), "bytes": os.path.getsize(path),
                                   "filters": {str(k): [int(x) for x in v] for k, v in d._filters.items()}}
        print(name, meta["cases"][name])
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
