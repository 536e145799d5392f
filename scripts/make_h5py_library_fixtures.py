"""Writes tests/golden/library_h5py_*.hdf5 with the REAL h5py / libhdf5 (this image: /opt/conda/bin/python3.9 has h5py
3.3.0 on libhdf5 1.10.6; the system python has neither).

    /opt/conda/bin/python3.9 scripts/make_h5py_library_fixtures.py

The calls are the ones the reference's library writer makes (ref: src/synference/library.py:4074-4153 -- `create_group("Grid")`,
`create_dataset(name, data=..., compression="gzip")`, list-of-str / str attributes on the root group, the FilterCodes-as-a-
dataset fallback of 4103-4110), so the files hold exactly the on-disk structures a real Synference library holds: chunked
(h5py's guessed chunk shape) + deflate datasets, variable-length UTF-8 string attributes, a v0 superblock with symbol-table
groups.  The expected arrays are regenerated from the seed by tests/test_cpu_hdf5_library.py (numpy's default_rng stream is
stable across the numpy versions involved, 1.26 here and 2.2 under test; the test also checks a checksum written below)."""
import json
import os
import sys

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def arrays(seed, C, D, N, n_supp):
    rng = np.random.default_rng(seed)
    phot = (rng.lognormal(size=(C, N)) * 100.0)
    par = rng.normal(size=(D, N))
    supp = rng.normal(size=(n_supp, N)) if n_supp else None
    return phot, par, supp


def write(name, seed, C, D, N, n_supp, *, codes_as_dataset=False, contiguous=False, shuffle=False, f32=False, libver=None):
    phot, par, supp = arrays(seed, C, D, N, n_supp)
    if f32:
        phot = phot.astype(np.float32)
    codes = [f"JWST/NIRCam.F{115 + 35 * i}W" for i in range(C)]
    names = ["log_mass", "tau_v", "log_zmet", "peak_age", "tau", "redshift", "xi", "beta"][:D]
    units = ["log10_Msun", "mag", "", "Myr", "dimensionless", "dimensionless", "", ""][:D]
    path = os.path.join(OUT, name)
    if os.path.exists(path):
        os.remove(path)
    kw = {} if libver is None else {"libver": libver}
    with h5py.File(path, "w", **kw) as f:
        g = f.create_group("Grid")
        dkw = {} if contiguous else {"compression": "gzip"}
        if shuffle:
            dkw["shuffle"] = True
        g.create_dataset("Photometry", data=phot, **dkw)
        g.create_dataset("Parameters", data=par, **dkw)
        if supp is not None:
            g.create_dataset("SupplementaryParameters", data=supp, **dkw)
        f.attrs["ParameterNames"] = names
        if codes_as_dataset:          # the reference's fallback when the attribute is too long (library.py:4103-4110)
            g.create_dataset("FilterCodes", data=np.array(codes, dtype="S"), compression="gzip")
            f.attrs["FilterCodes"] = "/Grid/FilterCodes/"
        else:
            f.attrs["FilterCodes"] = codes
        if supp is not None:
            f.attrs["SupplementaryParameterNames"] = ["mwa", "sfr_10"][:n_supp]
            f.attrs["SupplementaryParameterUnits"] = ["Myr", "Msun/yr"][:n_supp]
        f.attrs["PhotometryUnits"] = "nJy"
        f.attrs["ParameterUnits"] = units
        f.attrs["Grids"] = ["bpass-2.2.1-bin_chabrier03-0.1,300.0_cloudy-c23.01-sps"]
        f.attrs["CreationDT"] = "20260101_000000"
        g.create_dataset("redshift_grid", data=np.linspace(0.0, 12.0, 7), compression="gzip")   # "anything else as a dataset"
    meta = {"seed": seed, "C": C, "D": D, "N": N, "n_supp": n_supp, "codes_as_dataset": codes_as_dataset, "f32": f32,
            "phot_sum": float(np.asarray(phot, dtype=np.float64).sum()), "par_sum": float(par.sum()),
            "chunks": {k: (list(v.chunks) if v.chunks else None) for k, v in h5py.File(path, "r")["Grid"].items()},
            "bytes": os.path.getsize(path)}
    return meta


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {"h5py": h5py.__version__, "libhdf5": h5py.version.hdf5_version, "numpy": np.__version__, "python": sys.version.split()[0],
            "files": {}}
    meta["files"]["library_h5py_gzip.hdf5"] = write("library_h5py_gzip.hdf5", 11, 10, 5, 2500, 2)
    meta["files"]["library_h5py_codes_dataset.hdf5"] = write("library_h5py_codes_dataset.hdf5", 12, 20, 8, 900, 0, codes_as_dataset=True, shuffle=True)
    meta["files"]["library_h5py_contiguous_f32.hdf5"] = write("library_h5py_contiguous_f32.hdf5", 13, 4, 3, 333, 1, contiguous=True, f32=True)
    meta["files"]["library_h5py_latest.hdf5"] = write("library_h5py_latest.hdf5", 14, 6, 4, 700, 0, libver="latest")
    with open(os.path.join(OUT, "library_h5py_fixtures.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
