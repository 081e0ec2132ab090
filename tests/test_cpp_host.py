"""The compiled host layers over the C ABI: the C++ mirror (gogp_amd/host/gogp.hpp, plain g++)
and a plain C11 driver (tests/c_abi_driver.c, gcc -std=c11) that issues the calls in the order
the cgo shim does.  Both link libgogp_hip.so and reproduce reference known answers."""
import os
import subprocess

import pytest

from gogp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    _lib.build()
    exe = str(tmp_path / "cpp_host_driver")
    libdir = os.path.join(ROOT, "gogp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp_host_driver.cpp"),
                           "-o", exe, "-L" + libdir, "-lgogp_hip", "-Wl,-rpath," + libdir,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, r.stdout + r.stderr  # GOGP_EHIP: no CPU fallback


@pytest.mark.gpu
def test_cpp_host_known_answers(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp host ok" in r.stdout


def _build_c(tmp_path):
    _lib.build()
    exe = str(tmp_path / "c_abi_driver")
    libdir = os.path.join(ROOT, "gogp_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1",
                           os.path.join(ROOT, "tests", "c_abi_driver.c"), "-o", exe, "-L" + libdir,
                           "-lgogp_hip", "-lm", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_abi_driver_compiles_as_c11_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build_c(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, r.stdout + r.stderr  # GOGP_EHIP: no CPU fallback


@pytest.mark.gpu
def test_c_abi_driver_cgo_call_order(tmp_path):
    exe = _build_c(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c abi ok" in r.stdout
