"""ctypes view of oracle/_build/librbpf_oracle.so (the C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "librbpf_oracle.so")
_D = C.POINTER(C.c_double)
_LD = C.c_void_p  # long double arrays are passed as raw 16-byte-element buffers


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "rbpf_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE])
    lib = C.CDLL(LIB)
    lib.orc_map_new.restype = C.c_void_p; lib.orc_map_new.argtypes = [C.c_double, C.c_long]
    lib.orc_map_free.argtypes = [C.c_void_p]
    lib.orc_map_n_tiles.argtypes = [C.c_void_p]; lib.orc_map_dim.argtypes = [C.c_void_p]
    lib.orc_map_cells_visited.restype = C.c_ulonglong; lib.orc_map_cells_visited.argtypes = [C.c_void_p]
    lib.orc_map_tile.restype = _D; lib.orc_map_tile.argtypes = [C.c_void_p, C.c_int, _D]
    lib.orc_map_set_tile.argtypes = [C.c_void_p, C.c_long, C.c_long, _D]
    lib.orc_map_update.argtypes = [C.c_void_p, _LD, C.c_int, _D, _D, C.c_int]
    lib.orc_sample_weight.argtypes = [C.c_void_p, _LD, C.c_int, C.c_int, _D, _D, C.c_int, _D, _LD]
    lib.orc_robot_map_update.argtypes = [C.c_void_p, _LD, _LD, _LD, _D, _D, C.c_int, _D, _D, C.c_int]
    lib.orc_ld_to_pair.argtypes = [_LD, C.c_int, _D, _D]
    lib.orc_pair_to_ld.argtypes = [_D, _D, C.c_int, _LD]
    lib.orc_bench_particle_updates.restype = C.c_double
    lib.orc_bench_particle_updates.argtypes = [C.c_int, C.c_int, _D, _D, C.c_int, _D, _D, C.c_int, C.c_double]
    assert lib.orc_sizeof_long_double() == np.dtype(np.longdouble).itemsize == 16
    return lib


def dp(a):
    return a.ctypes.data_as(_D)


def ldp(a):
    assert a.dtype == np.longdouble and a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)


class CMap:
    def __init__(self, lib, cs=0.05, tile_len=40):
        self.lib, self.h = lib, lib.orc_map_new(cs, tile_len)
        self.dim = lib.orc_map_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_map_free(self.h); self.h = None

    def update(self, pose, sx, sy, ld=False):
        p = np.ascontiguousarray(pose, dtype=np.longdouble)
        sx, sy = np.ascontiguousarray(sx, dtype=np.float64), np.ascontiguousarray(sy, dtype=np.float64)
        self.lib.orc_map_update(self.h, ldp(p), int(ld), dp(sx), dp(sy), len(sx))

    def set_tile(self, cx, cy, cells):
        cells = np.ascontiguousarray(cells, dtype=np.float64)
        self.lib.orc_map_set_tile(self.h, int(cx), int(cy), dp(cells))

    def tiles(self):
        out = {}
        for k in range(self.lib.orc_map_n_tiles(self.h)):
            c = np.empty(2)
            ptr = self.lib.orc_map_tile(self.h, k, dp(c))
            out[(float(c[0]), float(c[1]))] = np.ctypeslib.as_array(ptr, shape=(self.dim, self.dim)).copy()
        return out

    def sample_weight(self, guesses, sx, sy, prs, ld=False):
        g = np.ascontiguousarray(guesses, dtype=np.longdouble)
        sx, sy = np.ascontiguousarray(sx, dtype=np.float64), np.ascontiguousarray(sy, dtype=np.float64)
        prs = np.ascontiguousarray(prs, dtype=np.float64)
        w = np.empty(len(g), dtype=np.longdouble)
        self.lib.orc_sample_weight(self.h, ldp(g), int(ld), len(g), dp(sx), dp(sy), len(sx), dp(prs), ldp(w))
        return w

    def robot_map_update(self, pose, cov, weight, guesses, prs, sx, sy):
        p = np.ascontiguousarray(pose, dtype=np.longdouble)
        c = np.ascontiguousarray(np.asarray(cov, dtype=np.longdouble).ravel())
        w = np.array([weight], dtype=np.longdouble)
        g = np.ascontiguousarray(guesses, dtype=np.float64); prs = np.ascontiguousarray(prs, dtype=np.float64)
        sx, sy = np.ascontiguousarray(sx, dtype=np.float64), np.ascontiguousarray(sy, dtype=np.float64)
        self.lib.orc_robot_map_update(self.h, ldp(p), ldp(c), ldp(w), dp(g), dp(prs), len(prs), dp(sx), dp(sy), len(sx))
        return p, c.reshape(3, 3), w[0]
