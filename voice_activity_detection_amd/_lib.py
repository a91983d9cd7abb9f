"""ctypes binding of libsavad.so (C ABI: include/savad.h).  Fails loudly: there is no CPU
fallback in the product path -- if the HIP library is missing or stale, importing it raises."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
import os

LIB_PATH = Path(os.environ.get("SAVAD_LIB", _PKG / "libsavad.so"))  # override only for kernel experiments


class SavadError(RuntimeError):
    pass


class savad_config(ctypes.Structure):
    _fields_ = [("feature_size", c_int32), ("num_layers", c_int32), ("d_model", c_int32)]


# every symbol include/savad.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "savad_create": (c_int, [POINTER(savad_config), POINTER(c_void_p)]),
    "savad_destroy": (None, [c_void_p]),
    "savad_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t, c_void_p]),
    "savad_num_params": (c_int, [c_void_p]),
    "savad_param_key": (c_char_p, [c_void_p, c_int]),
    "savad_param_numel": (c_size_t, [c_void_p, c_int]),
    "savad_reserve": (c_int, [c_void_p, c_int, c_void_p]),
    "savad_workspace_bytes": (c_int, [c_void_p, c_int, c_int, POINTER(c_size_t)]),
    "savad_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "savad_set_precision": (c_int, [c_void_p, c_int]),
    "savad_residual_saturations": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong), c_void_p]),
    "savad_forward_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "savad_forward_strided": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "savad_set_attention_splits": (c_int, [c_void_p, c_int]),
    "savad_set_row_mode": (c_int, [c_void_p, c_int]),
    "savad_set_batch_invariant": (c_int, [c_void_p, c_int]),
    "savad_set_profiling": (c_int, [c_void_p, c_int]),
    "savad_profiling_skip": (c_int, [c_void_p, c_int]),
    "savad_last_kernel_times": (c_int, [c_void_p, POINTER(c_char_p), POINTER(c_float), c_int]),
    "savad_window_offsets": (c_int, [c_int, c_int, POINTER(c_int32)]),
    "savad_gather_windows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "savad_boost": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "savad_predict_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "savad_predict_probabilities": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "savad_stream_window_count": (c_int, [c_int, c_int, c_int]),
    "savad_pcm16_to_f32": (c_int, [c_void_p, ctypes.c_long, c_void_p, c_void_p]),
    "savad_gather_strided": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "savad_overlap_merge": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "savad_logmel_frames": (c_int, [c_int]),
    "savad_logmel_workspace_bytes": (c_size_t, [c_int]),
    "savad_logmel": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "savad_logmel_span_samples": (c_int, [c_long, c_int, c_int, POINTER(c_long), POINTER(c_long)]),
    "savad_logmel_span_workspace_bytes": (c_size_t, [c_int]),
    "savad_logmel_span": (c_int, [c_void_p, c_long, c_long, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "savad_logmel_set_algorithm": (c_int, [c_int]),
    "savad_logmel_table_floats": (c_int, [c_int]),
    "savad_logmel_tables_host": (c_int, [c_void_p, c_void_p, c_void_p]),
    "savad_trim_voice_activity": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "savad_frames_to_samples": (c_long, [c_void_p, c_int, c_int, c_double, c_double, c_void_p]),
    "savad_samples_to_segments": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int]),
    "savad_optimal_split": (c_int, [c_void_p, c_void_p, c_long, c_long, c_void_p]),
    "savad_last_error": (c_char_p, []),
    "savad_version": (c_char_p, []),
}

_lib = None


def load():
    """Load libsavad.so (once).  `import torch` first so that the HIP runtime already mapped by
    torch (same SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps libamdhip64 before dlopen)

    if not LIB_PATH.exists():
        raise SavadError(
            f"{LIB_PATH} is missing: build it with `python -m voice_activity_detection_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise SavadError(f"libsavad error {rc}: {load().savad_last_error().decode()}")
