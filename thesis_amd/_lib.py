"""ctypes binding of librbpf_hip.so (C ABI in include/rbpf_hip.h).

The library is hand-written HIP for gfx950; it is the only compute path.  A missing
library is a hard error (no CPU fallback): build it with ``python -m thesis_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librbpf_hip.so")

RBPF_OK = 0
IMU_UNICYCLE, IMU_ABSOLUTE, IMU_VELOCITY = 0, 1, 2


class RbpfConfig(C.Structure):
    _fields_ = [
        ("n_particles", C.c_int32), ("n_samples", C.c_int32), ("max_beams", C.c_int32),
        ("tile_len_m", C.c_int32), ("cell_size", C.c_double), ("lattice_radius", C.c_int32),
        ("pool_tiles", C.c_int32), ("log_odds_occ", C.c_double), ("log_odds_nearby", C.c_double),
        ("max_odds_occ", C.c_double), ("log_odds_emp", C.c_double), ("min_odds_emp", C.c_double),
        ("quantum", C.c_double), ("occupied_threshold", C.c_double), ("max_ray_m", C.c_double),
        ("weight_min_range", C.c_double), ("weight_max_range", C.c_double),
        ("match_min_range", C.c_double), ("match_max_range", C.c_double),
        ("resample_spread", C.c_double), ("vel_noise", C.c_double * 4), ("device", C.c_int32),
        ("ndt_refine", C.c_int32), ("seed", C.c_uint64),
    ]


class RbpfCounters(C.Structure):
    _fields_ = [
        ("scan_updates", C.c_uint64), ("ray_cells_visited", C.c_uint64), ("cells_written", C.c_uint64),
        ("cells_gathered", C.c_uint64), ("tiles_in_use", C.c_uint64), ("resample_copies", C.c_uint64),
        ("bytes_copied", C.c_uint64), ("ms_raycast", C.c_double), ("ms_weight", C.c_double),
        ("ms_match", C.c_double), ("ms_resample", C.c_double), ("slow_cells", C.c_uint64),
        ("reserved", C.c_uint64 * 7), ("window_fallbacks", C.c_uint64), ("ndt_runs", C.c_uint64), ("ndt_evaluations", C.c_uint64),
        ("ndt_accepted", C.c_uint64), ("match_shared", C.c_uint64),
        ("fallback_reasons", C.c_uint64), ("ms_ndt", C.c_double), ("stamp7", C.c_uint64), ("map_windows", C.c_uint64),
        ("fallback_geometry", C.c_uint64), ("fallback_bound", C.c_uint64), ("fallback_tables", C.c_uint64),
        ("map_events", C.c_uint64), ("map_event_overflows", C.c_uint64),
    ]


_H = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)
_B = C.POINTER(C.c_int8)
_U = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); every symbol declared in include/rbpf_hip.h
PROTOTYPES = {
    "rbpf_default_config": (C.c_int, [C.POINTER(RbpfConfig)]),
    "rbpf_create": (C.c_int, [C.POINTER(RbpfConfig), C.POINTER(_H)]),
    "rbpf_destroy": (C.c_int, [_H]),
    "rbpf_last_error": (C.c_char_p, [_H]),
    "rbpf_set_stream": (C.c_int, [_H, C.c_void_p]),
    "rbpf_release_stream": (C.c_int, [_H]),
    "rbpf_abi_struct_bytes": (C.c_int, [_I, _I]),
    "rbpf_synchronize": (C.c_int, [_H]),
    "rbpf_get_counters": (C.c_int, [_H, C.POINTER(RbpfCounters)]),
    "rbpf_set_profiling": (C.c_int, [_H, C.c_int]),
    "rbpf_set_profiling_families": (C.c_int, [_H, C.c_uint32]),
    "rbpf_get_kernel_ms": (C.c_int, [_H, C.c_int32, _D, C.c_int32, _I]),
    "rbpf_set_scan": (C.c_int, [_H, _D, _D, C.c_int32]),
    "rbpf_set_scan_xy": (C.c_int, [_H, _D, _D, C.c_int32]),
    "rbpf_imu_update": (C.c_int, [_H, C.c_int32, _D, C.c_double]),
    "rbpf_weight_samples": (C.c_int, [_H, _D, _D, C.c_int32, _D]),
    "rbpf_map_update": (C.c_int, [_H, _D]),
    "rbpf_scan_update": (C.c_int, [_H, C.c_int32, _D, C.c_int32, _D, _D]),
    "rbpf_refresh_last_scan": (C.c_int, [_H, C.c_int32]),
    "rbpf_export_last_scan": (C.c_int, [_H, C.c_void_p, _I]),
    "rbpf_import_last_scan": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "rbpf_scan_update_begin": (C.c_int, [_H, C.c_int32, _D, C.c_int32, _D, _D]),
    "rbpf_scan_update_end": (C.c_int, [_H]),
    "rbpf_match_scan": (C.c_int, [_H, _D, C.c_int32, _D, C.c_int32, _D, C.c_int32, _D, _D, _D, _D]),
    "rbpf_match_inputs": (C.c_int, [_H, C.c_int32, _D, _D, _I, _D, _I, C.c_int32]),
    "rbpf_resample": (C.c_int, [_H, C.c_double, _I, _I]),
    "rbpf_export_weights": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "rbpf_resample_indices_global": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_double, _I, _I]),
    "rbpf_export_weights_early": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_void_p]),
    "rbpf_resample_indices_global_early": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "rbpf_resample_indices_global_wait": (C.c_int, [_H, _I, _I, _D]),
    "rbpf_apply_resample_local": (C.c_int, [_H, _I, _I]),
    "rbpf_set_global_ids": (C.c_int, [_H, _I]),
    "rbpf_pack_meta_width": (C.c_int32, [_H]),
    "rbpf_packed_particle_bytes": (C.c_int64, [_H]),
    "rbpf_pack_particles": (C.c_int, [_H, _I, C.c_int32, C.c_void_p, C.c_int64, _I, C.POINTER(C.c_int64)]),
    "rbpf_unpack_particles": (C.c_int, [_H, _I, C.c_int32, C.c_void_p, _I]),
    "rbpf_pack_raw_width": (C.c_int32, [_H]),
    "rbpf_gather_pack_meta": (C.c_int, [_H, _I, C.c_int32, C.c_void_p]),
    "rbpf_meta_from_raw": (C.c_int, [_H, _I, C.c_int32, _I, C.POINTER(C.c_int64)]),
    "rbpf_pack_particles_raw": (C.c_int, [_H, _I, C.c_int32, _I, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "rbpf_get_poses": (C.c_int, [_H, _D]),
    "rbpf_get_covs": (C.c_int, [_H, _D]),
    "rbpf_get_weights": (C.c_int, [_H, _D]),
    "rbpf_set_state": (C.c_int, [_H, _D, _D, _D]),
    "rbpf_get_rng_state": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rbpf_set_rng_state": (C.c_int, [_H, C.c_uint64, C.c_uint64]),
    "rbpf_get_tile_count": (C.c_int, [_H, C.c_int32, _I]),
    "rbpf_get_tile": (C.c_int, [_H, C.c_int32, C.c_int32, _D, _B]),
    "rbpf_set_tile": (C.c_int, [_H, C.c_int32, C.c_double, C.c_double, _B]),
    "rbpf_get_dim": (C.c_int, [_H, _I]),
    "rbpf_get_odds_at": (C.c_int, [_H, C.c_int32, _D, C.c_int32, _D, _U]),
}

_lib = None


def hip_runtime_paths():
    """Files of every libamdhip64 mapped into this process (there must never be two: see _share_torch_hip_runtime)."""
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return sorted(out)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  torch wheels for ROCm carry their own libamdhip64.so (soname libamdhip64.so.7, found
    through libtorch_hip's RPATH under the name libamdhip64.so), librbpf_hip.so is linked against /opt/rocm's
    libamdhip64.so.7.  If torch is imported first, the loader resolves our dependency to torch's copy by soname; if
    librbpf_hip.so comes first, /opt/rocm's copy is mapped and a later `import torch` maps a SECOND runtime, because
    the name it asks for matches neither the path nor the soname of the first.  Streams, events and device pointers
    that cross between torch and the engine (sharding.py: torch's current stream, RCCL buffers) are then handles of
    another runtime instance, and torch.cuda.synchronize() no longer waits for the engine's kernels.  So when torch is
    installed and no HIP runtime is mapped yet, torch's copy is loaded here first; torch finds the very same file later."""
    if hip_runtime_paths():
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


def load() -> C.CDLL:
    """Load librbpf_hip.so; raise if it is missing (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m thesis_amd.build`); thesis_amd has no CPU fallback")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    cb, kb = C.c_int32(), C.c_int32()
    fn = lib.rbpf_abi_struct_bytes
    fn.restype, fn.argtypes = C.c_int, [_I, _I]
    fn(C.byref(cb), C.byref(kb))
    if cb.value != C.sizeof(RbpfConfig) or kb.value != C.sizeof(RbpfCounters):
        raise ImportError(f"{LIB_PATH} was built from another include/rbpf_hip.h (struct sizes {cb.value}/{kb.value} vs "
                          f"{C.sizeof(RbpfConfig)}/{C.sizeof(RbpfCounters)}): run `python -m thesis_amd.build --force`")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
