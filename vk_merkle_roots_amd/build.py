"""Build recipes for the native parts of the engine (in-tree, gfx950 only).

`build_all()` is what `__graft_entry__.build()` runs.  hipcc cross-compiles for
gfx950 without a GPU.  Outputs land next to the package so that they travel to
the GPU box with the source snapshot:

    vk_merkle_roots_amd/libvkmr_hip.so   HIP kernels + the C ABI (include/vkmr_hip.h)
    vk_merkle_roots_amd/libvkmr_host.so  host-side helpers (stream packing, rndm generator)
    vk_merkle_roots_amd/bin/{vkmr,rndm,strm}  the C++ front end and its two feeder tools
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(CSRC, "host")
BIN = os.path.join(PKG, "bin")
HIP_LIB = os.path.join(PKG, "libvkmr_hip.so")
HOST_LIB = os.path.join(PKG, "libvkmr_host.so")
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


def _tree(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def build_hip(force=False):
    srcs = [os.path.join(CSRC, "vkmr_hip.hip")]
    deps = srcs + _tree(CSRC, (".hpp", ".h")) + [os.path.join(ROOT, "include", "vkmr_hip.h")]
    if not force and _newer(HIP_LIB, deps):
        return HIP_LIB
    _run([_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden",
          "-Wl,-rpath,/opt/rocm/lib", "-o", HIP_LIB] + srcs)
    return HIP_LIB


def build_host(force=False):
    """Host-side C++ (g++): helper library, front end, tools."""
    os.makedirs(BIN, exist_ok=True)
    hdrs = _tree(HOST, (".hpp", ".h")) + [os.path.join(ROOT, "include", "vkmr_hip.h")]
    cxx = os.environ.get("CXX", "g++")
    flags = ["-O2", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", HOST]
    built = []

    def need(target, srcs):
        return force or not _newer(target, srcs + hdrs)

    lib_srcs = [os.path.join(HOST, f) for f in ("stream_pack.cpp", "rndm_stream.cpp", "host_api.cpp", "cpu_sha256d.cpp") if os.path.exists(os.path.join(HOST, f))]
    if lib_srcs and need(HOST_LIB, lib_srcs):
        _run([cxx] + flags + ["-shared", "-fPIC", "-fvisibility=hidden", "-o", HOST_LIB] + lib_srcs)
    built.append(HOST_LIB)

    tools = {
        "rndm": ["rndm_main.cpp", "rndm_stream.cpp"],
        "strm": ["strm_main.cpp"],
    }
    for name, files in tools.items():
        srcs = [os.path.join(HOST, f) for f in files]
        if not all(os.path.exists(s) for s in srcs):
            continue
        out = os.path.join(BIN, name)
        if need(out, srcs):
            _run([cxx] + flags + ["-o", out] + srcs)
        built.append(out)

    vk_files = ["vkmr_main.cpp", "cpu_sha256d.cpp", "hip_sha256d.cpp", "inputs.cpp", "batches.cpp", "slices.cpp",
                "mappings.cpp", "reductions.cpp", "stream_pack.cpp"]
    vk_srcs = [os.path.join(HOST, f) for f in vk_files]
    if all(os.path.exists(s) for s in vk_srcs):
        out = os.path.join(BIN, "vkmr")
        if need(out, vk_srcs + [HIP_LIB]):
            _run([cxx] + flags + ["-o", out] + vk_srcs +
                 ["-L", PKG, "-lvkmr_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
        built.append(out)
    return built


def build_all(force=False):
    out = [build_hip(force)]
    out += build_host(force)
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv):
        print(p)
