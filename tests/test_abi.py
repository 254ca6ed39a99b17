"""CPU: the C-ABI library builds, loads and exports every symbol include/covo_hip.h declares
(no compute calls -- those need the GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from covo_mpc_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "covo_hip.h")).read()
    declared = set(re.findall(r"^(?:const char \*|int )\s*(covo_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 15
    lib = ctypes.CDLL(built.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in covo_hip.h but not exported"
    assert declared == set(built.EXPORTS), "python binding and header disagree"


def test_load_and_abi_version(built):
    lib = built.load_library()
    assert lib.covo_abi_version() == built.ABI_VERSION


def test_struct_layouts_match_header(built):
    # + rollover_terminate (ABI 2); + reward_kind, disturb_kind, disturb_period, disturb_scale, disturb_params[6], dyn_noise_scale (ABI 3)
    # + reset_traj, reserved0, reset_dt (double), reset_disturb_scale (double): the device env's auto-reset (ABI 6)
    assert ctypes.sizeof(built.EnvParamsC) == 4 * (1 + 3 + 3 + 5 + 1 + 1 + 1 + 4 + 6 + 1) + 4 + 4 + 8 + 8 == 128
    assert built.EnvParamsC.reset_traj.offset == 104 and built.EnvParamsC.reset_dt.offset == 112
    assert built.EnvParamsC.reward_kind.offset == 60 and built.EnvParamsC.disturb_params.offset == 76
    assert built.StepArgsC.rollout_deterministic.offset == built.StepArgsC.derive_keys.offset + 4
    assert ctypes.sizeof(built.ConfigC) == 24


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from covo_mpc_amd.controllers._core import SamplingCore
    with pytest.raises(built.CovoError):
        SamplingCore(1024, 32, 0.01, 1.0)


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "covo_mpc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "oracle/" not in src or f.endswith(".hip") or f.endswith(".hpp"), f
