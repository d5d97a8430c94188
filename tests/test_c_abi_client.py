"""include/rrt.h driven from plain C (tests/c_abi/c_abi_check.c): the header compiles as C99 and C++11, the
struct layouts are the ones the ctypes / Rust mirrors assume, and a furnace scene renders end to end."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c_abi", "c_abi_check.c")
BIN = os.path.join(ROOT, "tests", "c_abi", "c_abi_check")
CSRC = os.path.join(ROOT, "rustraytracer_amd", "csrc")


def build():
    lib = os.path.join(CSRC, "librrt_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", SRC, "-o", BIN, "-L" + CSRC, "-lrrt_hip",
                               "-lm", "-Wl,-rpath,$ORIGIN/../../rustraytracer_amd/csrc"])


def test_header_is_c_and_cxx():
    h = os.path.join(ROOT, "include", "rrt.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", h])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", h])


def test_c_client_host_side():
    build()
    r = subprocess.run([BIN, "cpu"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_client_renders():
    assert os.path.exists(BIN), "built on the CPU side (python -m pytest -m 'not gpu' or __graft_entry__.build())"
    r = subprocess.run([BIN, "gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
