"""Library (training set) files of the reference, read without h5py.

``load_library_from_hdf5`` has the signature and returns the dictionary of the reference's function of the same name
(ref: src/synference/utils.py:37-112): ``parameters`` (D, N), ``photometry`` (C, N) [nJy], ``filter_codes``,
``parameter_names``, ``parameter_units``, ``photometry_units`` and, when present, ``spectra`` and the supplementary
parameters.  The file is decoded by ``hdf5_lite`` (the structures h5py's default settings write,
ref: src/synference/library.py:4074-4153).
"""
from __future__ import annotations

import os

import numpy as np

from .hdf5_lite import File


def load_library_from_hdf5(hdf5_path: str, photometry_key: str = "Grid/Photometry", parameters_key: str = "Grid/Parameters",
                           filter_codes_attr: str = "FilterCodes", parameters_attr: str = "ParameterNames",
                           parameters_units_attr: str = "ParameterUnits", supp_key: str = "Grid/SupplementaryParameters",
                           supp_attr: str = "SupplementaryParameterNames", supp_units_attr: str = "SupplementaryParameterUnits",
                           phot_unit_attr: str = "PhotometryUnits", spectra_key: str = "Grid/Spectra",
                           pinned: bool = False, workers: "int | None" = None) -> dict:
    """``pinned``: the two big arrays (parameters, photometry) are decoded straight into page-locked host memory (numpy views
    of pinned torch tensors), so that the copy to the GPU that follows is one DMA each; ``workers``: chunk-inflating threads
    (hdf5_lite.Dataset.read).  Both are extensions of this backend; the returned dictionary is the reference's."""
    if not os.path.exists(hdf5_path):
        d = os.path.dirname(hdf5_path) or "."
        raise FileNotFoundError(f"HDF5 file not found: {hdf5_path}. Files in root directory: "
                                f"{os.listdir(d) if os.path.isdir(d) else []}")
    def big(ds):
        if not pinned:
            return ds.read(workers=workers)
        import torch
        buf = torch.empty(tuple(ds.shape), dtype=getattr(torch, str(np.dtype(ds.dtype.np_dtype).newbyteorder("=").name)),
                          pin_memory=torch.cuda.is_available())
        return ds.read(out=buf.numpy(), workers=workers)

    with File(hdf5_path) as f:
        parameters = big(f[parameters_key])
        filter_codes = f.attrs[filter_codes_attr]
        if isinstance(filter_codes, (bytes, str)):          # too long for an attribute: stored as a dataset (library.py:4103-4110)
            filter_codes = np.array([v.decode() if isinstance(v, bytes) else v
                                     for v in np.asarray(f[str(filter_codes).strip("/")][:]).reshape(-1).tolist()], dtype=object)
        output = {"parameters": parameters, "filter_codes": filter_codes, "parameter_names": f.attrs[parameters_attr],
                  "photometry_units": f.attrs[phot_unit_attr], "parameter_units": f.attrs.get(parameters_units_attr, None)}
        if photometry_key in f:
            output["photometry"] = big(f[photometry_key])
        if spectra_key in f:
            output["spectra"] = f[spectra_key][:]
        if supp_key in f:
            output["supplementary_parameters"] = f[supp_key][:]
            output["supplementary_parameter_names"] = f.attrs[supp_attr]
            output["supplementary_parameter_units"] = f.attrs[supp_units_attr]
    return output
