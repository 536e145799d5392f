"""Library files without h5py: synference_amd.hdf5_lite against byte-level fixtures assembled from
the HDF5 File Format Specification (tests/helpers/hdf5_fixture.py), and the reference's load_library_from_hdf5 contract
(ref: src/synference/utils.py:37-112)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
from hdf5_fixture import Writer, write_library  # noqa: E402

from synference_amd.hdf5_lite import File, Hdf5Error  # noqa: E402
from synference_amd.library import load_library_from_hdf5  # noqa: E402


@pytest.mark.parametrize("gzip,shuffle,chunks", [(4, False, None), (9, True, (3, 700)), (0, False, (10, 128)), (1, True, (1, 4096))])
def test_library_round_trip(tmp_path, gzip, shuffle, chunks):
    rng = np.random.default_rng(0)
    C, D, N = 10, 5, 3001                                        # ragged against every chunk shape
    phot = rng.lognormal(size=(C, N)) * 100.0
    par = rng.normal(size=(D, N))
    supp = rng.normal(size=(2, N))
    codes = [f"JWST/NIRCam.F{115 + 35 * i}W" for i in range(C)]
    names = ["log_mass", "tau_v", "log_zmet", "peak_age", "tau"]
    p = str(tmp_path / "grid.hdf5")
    write_library(p, phot, par, codes, names, ["Msun", "mag", "", "Myr", "dimensionless"], supp, ["mwa", "sfr"], ["Myr", "Msun/yr"],
                  chunks=chunks, gzip=gzip, shuffle=shuffle)
    out = load_library_from_hdf5(p)
    assert np.array_equal(out["photometry"], phot) and np.array_equal(out["parameters"], par)      # bit-exact
    assert out["photometry"].dtype == np.float64 and out["photometry"].shape == (C, N)
    assert list(out["filter_codes"]) == codes and list(out["parameter_names"]) == names
    assert out["photometry_units"] == "nJy" and list(out["parameter_units"])[-1] == "dimensionless"
    assert np.array_equal(out["supplementary_parameters"], supp) and list(out["supplementary_parameter_names"]) == ["mwa", "sfr"]
    assert "spectra" not in out
    with File(p) as f:
        assert "Grid/Photometry" in f and "Grid/Nope" not in f and sorted(f["Grid"].keys()) == ["Parameters", "Photometry", "SupplementaryParameters"]
        assert f.attrs["CreationDT"] == "20260101_000000"
        assert np.array_equal(f["Grid/Parameters"][:, 5:9], par[:, 5:9])


def test_contiguous_float32_dataset_fixed_strings_and_numeric_attributes(tmp_path):
    w = Writer()
    a = np.arange(24, dtype=np.float32).reshape(4, 6)
    d = w.dataset(a, chunks=None, attrs={"scale": np.array([1.5, 2.5]), "tags": np.array([b"ab", b"cde"], dtype="S3")})
    g, _, _ = w.group({"A": d}, attrs={"note": "inner"})
    p = str(tmp_path / "x.h5")
    w.finish({"G": g}, {"Title": "tést", "Names": ["a", "", "θ"]}, p)
    with File(p) as f:
        ds = f["G/A"]
        assert np.array_equal(ds[:], a) and ds[:].dtype == np.float32
        assert np.array_equal(ds.attrs["scale"], [1.5, 2.5]) and list(ds.attrs["tags"]) == ["ab", "cde"]
        assert f["G"].attrs["note"] == "inner" and f.attrs["Title"] == "tést" and list(f.attrs["Names"]) == ["a", "", "θ"]
        with pytest.raises(KeyError):
            f["G/B"]


def test_not_hdf5_and_missing_file(tmp_path):
    p = tmp_path / "junk.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(Hdf5Error):
        File(str(p))
    with pytest.raises(FileNotFoundError, match="HDF5 file not found"):
        load_library_from_hdf5(str(tmp_path / "absent.hdf5"))


def test_fitter_from_library_file(tmp_path):
    """SBI_Fitter.init_from_hdf5 (ref: sbi_runner.py:309-405): arrays in the reference's orientation."""
    from synference_amd import SBI_Fitter
    rng = np.random.default_rng(1)
    phot, par = rng.lognormal(size=(4, 50)), rng.normal(size=(3, 50))
    p = str(tmp_path / "lib.hdf5")
    write_library(p, phot, par, ["F1", "F2", "F3", "F4"], ["a", "b", "c"], ["", "", ""])
    f = SBI_Fitter.init_from_hdf5("m", p)
    assert f.fitted_parameter_array.shape == (50, 3) and np.array_equal(f.fitted_parameter_array, par.T)
    assert f.raw_observation_grid.shape == (4, 50) and f.raw_observation_names == ["F1", "F2", "F3", "F4"]
    assert f.parameter_names == ["a", "b", "c"] and f.library_path == p and not f.has_features
    out = SBI_Fitter.init_from_hdf5("m", p, return_output=True)
    assert set(out) >= {"parameters", "photometry", "filter_codes", "parameter_names", "photometry_units", "parameter_units"}


def test_large_chunked_library_is_inflated_on_a_thread_pool(tmp_path):
    """SURVEY 8f f1 at size: a chunked + deflated + shuffled library (2.5e5 galaxies x 20 filters here; the 1e6-row file of
    BASELINE configs[3] goes through the same path in tests/test_gpu_configs.py) read with several workers equals the
    single-thread read and the arrays that were written; the rate is printed."""
    import time
    from helpers.hdf5_fixture import write_library
    from synference_amd.hdf5_lite import File
    from synference_amd.library import load_library_from_hdf5
    rng = np.random.default_rng(3)
    N, C, D = 250_000, 20, 8
    phot = (rng.lognormal(2.0, 1.0, size=(C, N))).astype(np.float64)
    par = rng.normal(size=(D, N)).astype(np.float64)
    path = str(tmp_path / "big.h5")
    write_library(path, phot, par, [f"F{i}" for i in range(C)], [f"p{i}" for i in range(D)], chunks=(4, 8192), gzip=1, shuffle=True)
    t0 = time.perf_counter()
    lib = load_library_from_hdf5(path, workers=8)
    dt = time.perf_counter() - t0
    assert np.array_equal(lib["photometry"], phot) and np.array_equal(lib["parameters"], par)
    with File(path) as f:
        t0 = time.perf_counter()
        one = f["Grid/Photometry"].read(workers=1)
        dt1 = time.perf_counter() - t0
        buf = np.empty((C, N), dtype=np.float64)
        assert f["Grid/Photometry"].read(out=buf, workers=4) is buf
    assert np.array_equal(one, phot) and np.array_equal(buf, phot)
    print(f"hdf5_lite: {N} rows x ({C} + {D}) float64 in {dt:.2f} s with 8 workers = {N / dt / 1e6:.2f} M rows/s "
          f"(photometry alone, 1 worker: {dt1:.2f} s)")


# ---- files written by the real h5py / libhdf5 (scripts/make_h5py_library_fixtures.py, run with /opt/conda/bin/python3.9) --------
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _h5py_meta():
    import json
    with open(os.path.join(GOLDEN, "library_h5py_fixtures.json")) as fh:
        return json.load(fh)


def _expected(m):
    rng = np.random.default_rng(m["seed"])
    phot = rng.lognormal(size=(m["C"], m["N"])) * 100.0
    par = rng.normal(size=(m["D"], m["N"]))
    supp = rng.normal(size=(m["n_supp"], m["N"])) if m["n_supp"] else None
    if m["f32"]:
        phot = phot.astype(np.float32)
    return phot, par, supp


@pytest.mark.parametrize("name", ["library_h5py_gzip.hdf5", "library_h5py_codes_dataset.hdf5", "library_h5py_contiguous_f32.hdf5"])
def test_reads_libraries_written_by_h5py_like_the_reference_writer(name):
    """The reference writes its libraries with h5py's defaults + compression="gzip" (library.py:4074-4153): chunked with h5py's
    guessed chunk shape, deflate (here also + shuffle, contiguous float32, FilterCodes as a fixed-string dataset), vlen UTF-8
    attributes.  These files come from h5py 3.3.0 / libhdf5 1.10.6 -- not from this repository's own fixture writer."""
    meta = _h5py_meta()
    m = meta["files"][name]
    phot, par, supp = _expected(m)
    assert abs(float(np.asarray(phot, dtype=np.float64).sum()) - m["phot_sum"]) < 1e-6 * abs(m["phot_sum"])   # same random stream as the writer's numpy
    out = load_library_from_hdf5(os.path.join(GOLDEN, name))
    assert out["photometry"].dtype == phot.dtype and np.array_equal(out["photometry"], phot)
    assert out["parameters"].dtype == np.float64 and np.array_equal(out["parameters"], par)
    assert list(out["filter_codes"]) == [f"JWST/NIRCam.F{115 + 35 * i}W" for i in range(m["C"])]
    assert list(out["parameter_names"]) == ["log_mass", "tau_v", "log_zmet", "peak_age", "tau", "redshift", "xi", "beta"][:m["D"]]
    assert out["photometry_units"] == "nJy"
    assert list(out["parameter_units"]) == ["log10_Msun", "mag", "", "Myr", "dimensionless", "dimensionless", "", ""][:m["D"]]
    if supp is not None:
        assert np.array_equal(out["supplementary_parameters"], supp)
        assert list(out["supplementary_parameter_names"]) == ["mwa", "sfr_10"][:m["n_supp"]]
        assert list(out["supplementary_parameter_units"]) == ["Myr", "Msun/yr"][:m["n_supp"]]
    else:
        assert "supplementary_parameters" not in out
    with File(os.path.join(GOLDEN, name)) as f:
        assert np.allclose(f["Grid/redshift_grid"][:], np.linspace(0.0, 12.0, 7))
        assert f.attrs["CreationDT"] == "20260101_000000" and list(f.attrs["Grids"])[0].startswith("bpass-2.2.1")
        assert np.array_equal(f["Grid/Parameters"][:, 17:33], par[:, 17:33])
    # threads + caller-provided buffer (the path SBI_Fitter.init_from_hdf5 takes)
    out2 = load_library_from_hdf5(os.path.join(GOLDEN, name), workers=3)
    assert np.array_equal(out2["photometry"], phot)


def test_h5py_libver_latest_file():
    """h5py's libver="latest" (superblock v3, version-2 object headers, compact link storage, version-4 chunk layouts): NOT
    what the reference writes (it opens files with the default libver) -- either read correctly or refused with a clear error."""
    meta = _h5py_meta()
    m = meta["files"]["library_h5py_latest.hdf5"]
    phot, par, _ = _expected(m)
    try:
        out = load_library_from_hdf5(os.path.join(GOLDEN, "library_h5py_latest.hdf5"))
    except Hdf5Error as e:
        assert "not supported" in str(e) or "unsupported" in str(e).lower()
        return
    assert np.array_equal(out["photometry"], phot) and np.array_equal(out["parameters"], par)


def test_own_fixture_writer_is_valid_hdf5_according_to_libhdf5(tmp_path):
    """The byte-level fixture writer of this test-suite (tests/helpers/hdf5_fixture.py) judged by libhdf5's own tools: h5dump
    must parse the file and print the data it was given (skipped where the image has no h5dump)."""
    import shutil
    import subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if not h5dump:
        pytest.skip("no h5dump in this image")
    rng = np.random.default_rng(5)
    phot, par = rng.lognormal(size=(3, 50)), rng.normal(size=(2, 50))
    p = str(tmp_path / "own.hdf5")
    write_library(p, phot, par, ["a", "b", "c"], ["m", "z"], ["Msun", ""], None, None, None, chunks=(2, 32), gzip=4, shuffle=True)
    r = subprocess.run([h5dump, "-d", "/Grid/Parameters", "-m", "%.17g", p], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = [float(tok.rstrip(",")) for line in r.stdout.splitlines() if line.strip().startswith("(") for tok in line.split(":", 1)[1].split()]
    assert len(vals) == par.size and np.array_equal(np.array(vals).reshape(par.shape), par)
    r = subprocess.run([h5dump, "-p", "-H", p], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and '"FilterCodes"' in r.stdout and "SHUFFLE" in r.stdout and "DEFLATE" in r.stdout, r.stdout + r.stderr
