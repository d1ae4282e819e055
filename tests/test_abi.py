"""The C-ABI library loads and exports every symbol include/rbpf_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from thesis_amd import build, _lib
    build.build_extension(verbose=False)          # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load()


def header_symbols():
    src = open(os.path.join(REPO, "include", "rbpf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rbpf_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from thesis_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rbpf_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, "ctypes prototypes and header disagree"


def test_config_struct_layout(lib):
    from thesis_amd._lib import RbpfConfig
    cfg = RbpfConfig()
    assert lib.rbpf_default_config(ctypes.byref(cfg)) == 0
    # reference constants: robot.py:17, hybridmap.py:67-68, gridmap.py:20-24, main.py:50
    assert (cfg.n_samples, cfg.tile_len_m, cfg.cell_size) == (30, 40, 0.05)
    assert (cfg.log_odds_occ, cfg.log_odds_nearby, cfg.max_odds_occ, cfg.log_odds_emp, cfg.min_odds_emp) == (0.8, 0.2, 3.0, -0.3, -3.0)
    assert cfg.resample_spread == 200.0 and cfg.max_ray_m == 15.0 and cfg.seed == 42
    assert list(cfg.vel_noise) == [0.02, 0.01, 0.2, 0.02]


def test_bad_config_is_rejected_before_touching_the_gpu(lib):
    from thesis_amd._lib import RbpfConfig
    cfg = RbpfConfig()
    lib.rbpf_default_config(ctypes.byref(cfg))
    cfg.log_odds_occ = 0.85                      # not a multiple of the 0.1 quantum
    h = ctypes.c_void_p()
    assert lib.rbpf_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"quantum" in lib.rbpf_last_error(None)
