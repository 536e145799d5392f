"""ctypes binding of include/synference_hip.h.

The library is the product: there is NO CPU fallback.  ``load()`` raises if the shared
object is missing, and every compute entry point raises ``RuntimeError`` (with
``sf_last_error()``) when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libsynference_hip.so"

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)


class sf_flow_desc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("D", C.c_int32), ("C", C.c_int32), ("H", C.c_int32),
        ("T", C.c_int32), ("K", C.c_int32), ("NB", C.c_int32), ("scale_fn", C.c_int32),
        ("hidden_bf16", C.c_int32),
        ("tail_bound", C.c_float), ("min_bin_width", C.c_float), ("min_bin_height", C.c_float),
        ("min_derivative", C.c_float), ("maf_eps", C.c_float), ("lu_eps", C.c_float),
        ("theta_mean", c_f32p), ("theta_std", c_f32p), ("x_mean", c_f32p), ("x_std", c_f32p),
        ("perms", c_i32p), ("ar_slope", C.c_float),
    ]


class sf_mlp_desc(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("n_layers", C.c_int32), ("widths", C.c_int32 * 4), ("act", C.c_int32),
                ("x_mean", c_f32p), ("x_std", c_f32p)]


class sf_adam_desc(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("decoupled", C.c_int32)]


# name -> (restype, argtypes); mirrors include/synference_hip.h one for one
PROTOTYPES = {
    "sf_flow_create": (C.c_int, [C.POINTER(sf_flow_desc), C.POINTER(C.c_void_p)]),
    "sf_flow_destroy": (None, [C.c_void_p]),
    "sf_flow_num_params": (C.c_int64, [C.c_void_p]),
    "sf_flow_packed_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_set_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "sf_flow_pack_table": (C.c_int, [C.c_void_p, c_i32p, c_i32p, C.c_int64]),
    "sf_flow_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "sf_flow_log_prob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sf_flow_inverse_from_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "sf_flow_sample_round": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_uint32, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_flux_to_asinh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "sf_scatter_depths": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_pit_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "sf_copy_to_host_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sf_flow_packed16_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_pack_table16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sf_flow_packed16b_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_inverse_from_noise_sampler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sf_set_sampler_fp32": (C.c_int, [C.c_int]),
    "sf_flow_set_sample_row_offset": (C.c_int, [C.c_void_p, C.c_int64]),
    "sf_flow_train_path": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "sf_flow_trainc_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_trainc_grad_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_cst_size": (C.c_int64, [C.c_void_p]),
    "sf_flow_trainc_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_int64]),
    "sf_flow_pack_table16b": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sf_flow_get_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "sf_flow_prepare_context": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sf_flow_release_context": (C.c_int, [C.c_void_p]),
    "sf_flow_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                 C.c_void_p]),
    "sf_flow_sample_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "sf_flow_sample_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sf_flow_acceptance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_void_p]),
    "sf_flow_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_flow_loss_grad_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "sf_flow_train_epoch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                      C.c_float, C.c_void_p, C.c_void_p, C.POINTER(sf_adam_desc), C.c_int64, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_flow_train_epoch_dp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                         C.c_float, C.c_void_p, C.c_void_p, C.POINTER(sf_adam_desc), C.c_int64, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_comm_set_library": (C.c_int, [C.c_char_p]),
    "sf_comm_library": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]),
    "sf_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int64]),
    "sf_comm_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sf_comm_destroy": (None, [C.c_void_p]),
    "sf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_comm_all_reduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sf_flow_loss_grad_weighted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "sf_flow_set_sample_time_limit": (C.c_int, [C.c_void_p, C.c_double]),
    "sf_flow_set_sample_output_f64": (C.c_int, [C.c_void_p, C.c_int]),
    "sf_flow_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "sf_flow_train_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sf_opt_create": (C.c_int, [C.c_int64, C.POINTER(sf_adam_desc), C.POINTER(C.c_void_p)]),
    "sf_opt_destroy": (None, [C.c_void_p]),
    "sf_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "sf_adam_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.POINTER(sf_adam_desc), C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "sf_opt_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.POINTER(C.c_int64))]),
    "sf_mlp_create": (C.c_int, [C.POINTER(sf_mlp_desc), C.POINTER(C.c_void_p)]),
    "sf_mlp_destroy": (None, [C.c_void_p]),
    "sf_mlp_num_params": (C.c_int64, [C.c_void_p]),
    "sf_mlp_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sf_mlp_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p]),
    "sf_quantiles": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                               C.c_void_p]),
    "sf_flux_to_abmag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_last_error": (C.c_char_p, []),
    "sf_version": (C.c_char_p, []),
    "sf_device_count": (C.c_int, []),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library (built by ``__graft_entry__.build()`` / ``make -C synference_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("SYNFERENCE_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the flow engine.")
    lib = C.CDLL(str(path))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().sf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"synference_hip error {rc}: {msg}")
