"""The host-side surface-area tree builder (rustraytracer_amd/csrc/rrt_sah.hpp) on the CPU: a C++ harness checks
that every primitive sits in exactly one leaf, that child boxes contain their primitives, that leaves respect the
size limit, and that the tree does not depend on the number of threads.  (That rays find the same hits through
this tree as through the scan is checked on the GPU, tests/test_gpu_lbvh.py.)"""
import os
import subprocess

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "sah_check.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "sah_check")
CSRC = os.path.join(ROOT, "rustraytracer_amd", "csrc")


def test_sah_builder_invariants():
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread", "-I" + CSRC, SRC, "-o", BIN])
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")
