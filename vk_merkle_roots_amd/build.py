"""Build recipes for the native parts of the engine (in-tree, gfx950 only).

`build_all()` is what `__graft_entry__.build()` runs.  hipcc cross-compiles for
gfx950 without a GPU.  Outputs land next to the package so that they travel to
the GPU box with the source snapshot:

    vk_merkle_roots_amd/libvkmr_hip.so   HIP kernels + the C ABI (include/vkmr_hip.h)
    vk_merkle_roots_amd/libvkmr_hip_stamps.so   DIAGNOSTIC build of the same source with -DVKMR_STAMPS: in-kernel
                                         s_memtime / s_memrealtime stamps (tools/kernel_clock.py, bench.py's
                                         measured shader clock).  Never loaded by the product.
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
PIPELINE_LIB = os.path.join(PKG, "libvkmr_pipeline.so")   # the C++ stream processor behind one C entry point (needs libvkmr_hip.so)
STAMPS_LIB = os.path.join(PKG, "libvkmr_hip_stamps.so")
EXP_LIB = os.path.join(ROOT, "build", "ab", "libexp.so")
ARCH = "gfx950"
SPLIT_ADD3_EVERY = 4   # isa_prio_pass: every 4th v_add3_u32 becomes two v_add_u32 (balances the two issue slots: -1.4 %, profiles/r03_ab_add3_split.txt)
LATENCY_BOUND_KERNELS = ("reduce_collapse_kernel", "reduce_tail_kernel", "reduce_collapse_proofs_kernel", "reduce_tail_proofs_kernel")   # one wavefront per SIMD: left as hipcc emits them (collapse: 113 vs 140 us with the pass, profiles/r03_reduce_top_kernels.txt)
ROTATE_LEVEL = None    # isa_prio_pass: priority of v_alignbit_b32 when it differs from the other complex instructions' (1)
PRIO_GAP = 0    # isa_prio_pass: complex-instruction runs separated by at most this many simple instructions are merged


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


def _llvm(tool):
    for d in (os.environ.get("ROCM_LLVM_BIN"), "/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"):
        if d and os.path.exists(os.path.join(d, tool)):
            return os.path.join(d, tool)
    raise RuntimeError(f"{tool} not found under /opt/rocm/lib/llvm/bin")


def source_id(defines=(), prio_gap=PRIO_GAP, split_every=SPLIT_ADD3_EVERY):
    """Identity of what a kernel library is made from: the device sources, the C ABI header, the issue pass and the
    build parameters.  Compiled into the library (vkmr_hip_kernel_info: " build=<id>") and recorded beside every
    profile (profiles/pmc_latest.json), so that a counter record is only ever quoted for the kernels that produced it."""
    import zlib
    crc, adl = 0, 1      # two independent 32-bit checksums over the same byte stream: 16 hex digits of identity
    def feed(b):
        nonlocal crc, adl
        crc, adl = zlib.crc32(b, crc), zlib.adler32(b, adl)
    files = [os.path.join(CSRC, "vkmr_hip.hip")] + _tree(CSRC, (".hpp", ".h")) + [os.path.join(ROOT, "include", "vkmr_hip.h"),
                                                                                     os.path.join(PKG, "isa_prio_pass.py")]
    for f in sorted(files):
        if os.sep + "host" + os.sep in f:
            continue            # host-side C++ is not in the kernel library
        if "experiments" in os.path.relpath(f, CSRC) and "-DVKMR_EXPERIMENTS" not in defines:
            continue            # map_experiments.hpp, experiments/*: compiled into the experiments build only
        feed(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            feed(fh.read())
    feed(repr((sorted(defines), prio_gap, split_every, LATENCY_BOUND_KERNELS, ARCH)).encode())
    return f"{crc & 0xffffffff:08x}{adl & 0xffffffff:08x}"


def _build_hip_variant(target, defines, force, prio_gap=PRIO_GAP, prio_level=1, split_every=SPLIT_ADD3_EVERY, rotate_level=ROTATE_LEVEL):
    """hipcc in five explicit steps so that the issue-priority pass (isa_prio_pass.py) can run on the device assembly:
    device code -> .s, pass, assemble + link the code object, bundle it, compile the host side around that bundle.
    prio_gap None = plain one-step hipcc build (no pass), for A/B timing."""
    src = os.path.join(CSRC, "vkmr_hip.hip")
    deps = [src, os.path.join(PKG, "isa_prio_pass.py"), os.path.abspath(__file__)] + _tree(CSRC, (".hpp", ".h")) + [os.path.join(ROOT, "include", "vkmr_hip.h")]
    if not force and _newer(target, deps):
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    common = (["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden"] + defines +
              [f'-DVKMR_BUILD_ID="{source_id(defines, prio_gap, split_every)}"'])
    if prio_gap is None:
        _run([_hipcc()] + common + ["-shared", "-Wl,-rpath,/opt/rocm/lib", "-o", target, src])
        return target
    try:
        for tool in ("clang", "lld", "clang-offload-bundler"):
            _llvm(tool)
    except RuntimeError as e:
        # the issue pass needs the llvm tools beside hipcc: without them the library is still built -- the plain one-step way,
        # under a build id of its own (no pass: 25 % slower kernels, no static counts) -- and says so
        sys.stderr.write(f"[build] WARNING: {e}; building {os.path.basename(target)} WITHOUT the issue-priority pass\n")
        common = [c for c in common if not c.startswith("-DVKMR_BUILD_ID=")] + [f'-DVKMR_BUILD_ID="{source_id(defines, None, split_every)}"']
        _run([_hipcc()] + common + ["-shared", "-Wl,-rpath,/opt/rocm/lib", "-o", target, src])
        isa = os.path.splitext(target)[0] + ".isa.json"
        if os.path.exists(isa):
            os.remove(isa)          # counts of another build must not sit beside this one
        return target
    work = os.path.join(ROOT, "build", "obj", os.path.basename(target))
    os.makedirs(work, exist_ok=True)
    dev_s, prio_s = os.path.join(work, "device.s"), os.path.join(work, "device_prio.s")
    dev_o, dev_co, fatbin, host_o = (os.path.join(work, n) for n in ("device.o", "device.co", "device.hipfb", "host.o"))
    _run([_hipcc()] + common + ["--cuda-device-only", "-S", "-o", dev_s, src])
    from . import isa_prio_pass
    with open(dev_s) as f:
        lines = f.readlines()
    out, stats = isa_prio_pass.transform(lines, prio_gap, prio_level, split_every, LATENCY_BOUND_KERNELS, rotate_level)
    with open(prio_s, "w") as f:
        f.writelines(out)
    # the pass rewrites compiler output with regular expressions: check that it did what it says and nothing else, and that
    # every opcode the static counts will price is one the issue measurements covered, BEFORE anything is assembled
    diffs = isa_prio_pass.verify(lines, out)
    if diffs:
        raise RuntimeError("isa_prio_pass changed more than s_setprio insertions and add3 splits:\n  " + "\n  ".join(diffs))
    audit = isa_prio_pass.audit(out)
    if audit["unclassified"] or audit["block_count_errors"]:
        sys.stderr.write("[build] WARNING: the issue model does not cover this listing (tests/test_isa_prio_pass.py will fail):\n  " +
                         "\n  ".join(audit["unclassified"] + audit["block_count_errors"]) + "\n")
    _run([_llvm("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", prio_s, "-o", dev_o])
    _run([_llvm("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_co, dev_o])
    _run([_llvm("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={dev_co}", f"-output={fatbin}"])
    _run([_hipcc()] + common + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-o", host_o])
    _run([_hipcc(), "-shared", "-Wl,-rpath,/opt/rocm/lib", "-o", target, host_o])
    with open(os.path.join(work, "prio_pass_stats.txt"), "w") as f:
        f.write(repr(stats) + "\n")
    # static instruction counts of the hash blocks, for bench.py's issue roofline; travels with the library
    import json
    info = {"build": source_id(defines, prio_gap, split_every), "library": os.path.basename(target), "prio_gap": prio_gap,
            "split_add3_every": split_every, "pass": stats, "hash_blocks": isa_prio_pass.hash_blocks(out), "audit": audit,
            "verified": "output == input + s_setprio insertions + add3 splits (isa_prio_pass.verify)"}
    with open(os.path.splitext(target)[0] + ".isa.json", "w") as f:
        json.dump(info, f, indent=1)
    return target


def build_hip(force=False):
    """The product library: no experiment knobs, no stamps."""
    return _build_hip_variant(HIP_LIB, [], force)


def build_stamps(force=False):
    """Diagnostic twin of the product library (same source, -DVKMR_STAMPS): measures the shader clock the chip
    holds inside map_kernel / reduce_pass_kernel.  Loaded only through VKMR_HIP_LIB by tools and bench.py's clock leg."""
    return _build_hip_variant(STAMPS_LIB, ["-DVKMR_STAMPS"], force)


def build_experiments(force=False):
    """Tools build (-DVKMR_EXPERIMENTS): the A/B knobs VKMR_MAP_VARIANT/_FIT/_TILE/_DYNLDS and the non-shipped
    map_kernel instantiations (csrc/map_experiments.hpp).  Lands under build/ab/, not in the package."""
    return _build_hip_variant(EXP_LIB, ["-DVKMR_EXPERIMENTS"], force)


def build_experiments_stamps(force=False):
    """The experiments build with the in-kernel stamps as well (tools/kernel_clock.py --lib build/ab/libexp_stamps.so with
    VKMR_MAP_VARIANT set): clock and phase shares of a variant that is not the shipped one."""
    return _build_hip_variant(os.path.join(ROOT, "build", "ab", "libexp_stamps.so"), ["-DVKMR_EXPERIMENTS", "-DVKMR_STAMPS"], force)


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

    pipe_files = ["packed_pipeline.cpp", "cpu_sha256d.cpp", "hip_sha256d.cpp", "batches.cpp", "slices.cpp", "mappings.cpp", "reductions.cpp",
                  "stream_pack.cpp"]
    pipe_srcs = [os.path.join(HOST, f) for f in pipe_files]
    if all(os.path.exists(s) for s in pipe_srcs):
        if need(PIPELINE_LIB, pipe_srcs + [HIP_LIB]):
            _run([cxx] + flags + ["-shared", "-fPIC", "-fvisibility=hidden", "-o", PIPELINE_LIB] + pipe_srcs +
                 ["-L", PKG, "-lvkmr_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
        built.append(PIPELINE_LIB)

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
    out = [build_hip(force), build_stamps(force), build_experiments(force)]
    out += build_host(force)
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv):
        print(p)

