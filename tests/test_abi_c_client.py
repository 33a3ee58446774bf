"""A plain C99 program against include/vkmr_hip.h + libvkmr_hip.so: compiles everywhere, runs on the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT


def build_client(native, tmp_path):
    exe = str(tmp_path / "abi_client")
    libdir = os.path.dirname(native.HIP_LIB)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", exe,
                           "-L", libdir, "-lvkmr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_client_compiles_and_links(native, tmp_path):
    exe = build_client(native, tmp_path)
    r = subprocess.run([exe, "a"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    # without a GPU it reports "no HIP device" (exit 3); with one it prints a root
    assert r.returncode in (0, 3), r.stderr


@pytest.mark.gpu
def test_c_client_roots(native, oracle, tmp_path):
    exe = build_client(native, tmp_path)
    for strings in (["a"], ["a", "b"], ["a", "b", "c"], ["%02d" % i + "x" * (i * 7 % 130) for i in range(100)]):
        r = subprocess.run([exe] + strings, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        want, cnt, _ = oracle.root_of_stream("\n".join(strings).encode())
        assert cnt == len(strings) and r.stdout.decode().strip() == want
