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


def test_struct_sizes_match_the_library(lib):
    """The ctypes mirrors of rbpf_config / rbpf_counters have the layout the library was compiled with."""
    from thesis_amd._lib import RbpfConfig, RbpfCounters
    cb, kb = ctypes.c_int32(), ctypes.c_int32()
    assert lib.rbpf_abi_struct_bytes(ctypes.byref(cb), ctypes.byref(kb)) == 0
    assert (cb.value, kb.value) == (ctypes.sizeof(RbpfConfig), ctypes.sizeof(RbpfCounters))
    hdr = open(os.path.join(REPO, "include", "rbpf_hip.h")).read()
    body = hdr[hdr.index("typedef struct rbpf_counters {"):hdr.index("} rbpf_counters;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:uint64_t|double)\s+([a-z_0-9]+)(?:\[\d+\])?;", body)
    assert names == [n for n, _ in RbpfCounters._fields_], "rbpf_counters fields: header and ctypes mirror disagree"


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """librbpf_hip.so first, torch second (the order of every test module here): both must share ONE libamdhip64,
    or torch's streams / RCCL buffers would be handles of another runtime instance (thesis_amd/_lib.py)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from thesis_amd import _lib\n"
            "_lib.load()\n"
            "a = _lib.hip_runtime_paths()\n"
            "import torch\n"
            "b = _lib.hip_runtime_paths()\n"
            "print(len(a), len(b), a == b)\n") % REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-3:] == ["1", "1", "True"], out.stdout
