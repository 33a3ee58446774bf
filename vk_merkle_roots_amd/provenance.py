"""Which build a measurement belongs to.

The kernel library carries the identity of its sources and build parameters (build.source_id, compiled in as
" build=<id>" at the end of vkmr_hip_kernel_info()).  Counter records made from a profiled run
(profiles/pmc_latest.json, tools/pmc_to_json.py) store the id and the kernel instantiation they were measured on;
bench.py quotes their HBM traffic only for a library with the same id that launched the same kernel -- change a kernel
header and the figure is gone until the counters are collected again (`roofline.traffic` must not be a
replay of an older kernel's number)."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id_of(kernel_info):
    """The id inside a vkmr_hip_kernel_info() string; None when the library predates it."""
    m = re.search(r"\bbuild=([0-9a-f]{8,})", kernel_info or "")
    return m.group(1) if m else None


def map_symbol_of(kernel_info):
    """The map_kernel instantiation the last launch used, e.g. 'map_kernel<512, 1024, 17664, 0, false>'."""
    m = re.search(r"map=\(?(map_kernel<[^>]*>)", kernel_info or "")
    return m.group(1) if m else None


def commit_of(path):
    """Short hash of the last commit that touched `path` (None outside a git checkout: the GPU box has no .git)."""
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
        return r.stdout.decode().strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def traffic_from_pmc(record, kernel_info, field, **must_match):
    """(bytes or None, source dict).  `record`: parsed profiles/pmc_latest.json; `field`: the byte count wanted;
    must_match: workload keys (strings_per_map_launch, maxlen, slice_log2 ...) that have to agree as well."""
    src = {"file": "profiles/pmc_latest.json", "record_build": (record or {}).get("build"), "library_build": build_id_of(kernel_info),
           "record_map_kernel": (record or {}).get("map_kernel_symbol"), "library_map_kernel": map_symbol_of(kernel_info)}
    if not record or record.get(field) is None:
        return None, dict(src, used=False, why="no record")
    if src["library_build"] is None or src["record_build"] != src["library_build"]:
        return None, dict(src, used=False, why="the record was measured on a different build of the kernels")
    if field.startswith("map") and src["record_map_kernel"] != src["library_map_kernel"]:
        return None, dict(src, used=False, why="the record was measured on a different map_kernel instantiation")
    for k, v in must_match.items():
        if record.get(k) != v:
            return None, dict(src, used=False, why=f"workload differs ({k}: {record.get(k)!r} vs {v!r})")
    return record[field], dict(src, used=True, commit=commit_of("profiles/pmc_latest.json"))


def load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None
