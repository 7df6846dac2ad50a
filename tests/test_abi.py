"""CPU-side checks of the C-ABI boundary: the library builds/loads here and exports every symbol that
include/ddpm_hip.h declares (no compute calls - there is no GPU in this container)."""
import ctypes
import os
import re

import dm_amd_bootstrap  # noqa: F401
from dm_amd.runtime import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ddpm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddpm_\w+)\s*\(", src)))


def test_header_symbols_exported():
    assert os.path.exists(hip.LIB_PATH), "run python diffusion-model_amd/build.py"
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_header():
    names = set(declared_functions()) - {"ddpm_last_error"}
    assert names == set(hip._PROTOS), (names ^ set(hip._PROTOS))
    assert hip.lib().ddpm_abi_version() == 4


def test_invalid_arguments_are_reported_without_a_gpu():
    # argument validation happens before any HIP call, so it is testable on CPU
    rc = hip.lib().ddpm_q_sample(None, None, None, None, None, None, 0, 0, 0, 0, None, 0, None)
    assert rc != 0 and b"null" in hip.lib().ddpm_last_error()
