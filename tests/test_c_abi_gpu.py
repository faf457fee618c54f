"""Builds and runs tests/c_abi_example.c — a plain C program (HIP runtime only) calling libmultike_hip.so."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_caller(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_example")
    lib_dir = os.path.join(ROOT, "multike_amd")
    subprocess.check_call([hipcc, "-x", "hip", os.path.join(ROOT, "tests", "c_abi_example.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", lib_dir, "-lmultike_hip", f"-Wl,-rpath,{lib_dir}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI example ok" in out.stdout
