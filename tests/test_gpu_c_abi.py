"""The C ABI from plain C: tests/c/abi_smoke.c is compiled with gcc against include/m2s.h, linked with
libm2s_hip.so and run as its own process (no Python, no torch in that process).  Run with `-m gpu`."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, cc="gcc", std="-std=c99"):
    exe = str(tmp_path / "abi_smoke")
    cmd = [cc, std, "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-L", os.path.join(ROOT, "mesh_to_sdf_amd"), "-lm2s_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.join(ROOT, "mesh_to_sdf_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_header_compiles_as_c99_and_links(tmp_path):
    """No GPU needed: the header is valid C99 and every symbol the program uses resolves."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_c_program_runs_the_known_answers(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "sdf.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "FAIL" not in r.stdout


def build_cpp(tmp_path):
    exe = str(tmp_path / "reference_tests")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-L", os.path.join(ROOT, "mesh_to_sdf_amd"), "-lm2s_hip",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "mesh_to_sdf_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles(tmp_path):
    """include/mesh_to_sdf.hpp (the C++ mirror of the reference's interface) is valid C++17 and links."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    assert os.path.exists(build_cpp(tmp_path))


@pytest.mark.gpu
def test_reference_tests_in_cpp(tmp_path):
    """The reference crate's own test assertions, restated against the C++ mirror, in a process without Python."""
    exe = build_cpp(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and " 0 failed" in r.stdout, r.stdout + r.stderr
