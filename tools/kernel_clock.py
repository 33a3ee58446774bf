#!/usr/bin/env python3
"""The shader clock the chip HOLDS inside the shipped kernels, measured in-kernel.

Loads the diagnostic twin of the product library (vk_merkle_roots_amd/libvkmr_hip_stamps.so: the same source built
with -DVKMR_STAMPS, nothing else changed).  In it lane 0 of every workgroup's first wavefront stamps s_memtime (shader
cycles) and s_memrealtime (constant 100 MHz) at its first and last instruction into a buffer no kernel reads; the clock
of a workgroup is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6: after >= 2 s of
back-to-back launches on random data; s_memtime counters are per XCD, so only per-workgroup differences are used).
Board power and the driver's sclk are sampled from sysfs hwmon beside it.  Prints one JSON object (last line).

    python3 tools/kernel_clock.py [--leaves-log2 24] [--maxlen 127] [--seconds 2.0]        # GPU box only
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vk_merkle_roots_amd.build import STAMPS_LIB  # noqa: E402

if "--lib" in sys.argv:      # another stamped build (e.g. -DVKMR_STAMPS -DVKMR_EXPERIMENTS, with VKMR_MAP_VARIANT in the environment)
    STAMPS_LIB = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
os.environ["VKMR_HIP_LIB"] = STAMPS_LIB
import vk_merkle_roots_amd as vk  # noqa: E402

SLOTS = 65536


def _first(pattern):
    g = sorted(glob.glob(pattern))
    return g[0] if g else None


class Hwmon:
    """Board power (W) and the driver's shader clock (MHz) from sysfs; every field may be missing."""

    def __init__(self, pci_bus_id=None):
        # the box's host has several GPUs: take the hwmon node of the PCI function HIP device 0 sits on
        base = f"/sys/bus/pci/devices/{pci_bus_id.lower()}/hwmon/hwmon*/" if pci_bus_id else "/sys/class/drm/card*/device/hwmon/hwmon*/"
        self.base = base
        self.power = _first(base + "power1_average") or _first(base + "power1_input")
        self.cap = _first(base + "power1_cap")
        self.sclk = _first(base + "freq1_input")
        self.temps = {}                       # label -> path: junction / edge / memory temperatures where the driver shows them
        for t in sorted(glob.glob(base + "temp*_input")):
            label = t.replace("_input", "_label")
            try:
                with open(label) as f:
                    self.temps[f.read().strip()] = t
            except OSError:
                self.temps[os.path.basename(t)] = t

    @staticmethod
    def read(path, scale):
        try:
            with open(path) as f:
                return float(f.read().split()[0]) / scale
        except Exception:
            return None


def sustained(dev, launch, seconds, hw):
    """`seconds` of back-to-back launches with hwmon sampled from a side thread; returns (ms per launch, power W, sclk MHz)."""
    stop, pw, sk = threading.Event(), [], []
    hw.last_temps = {}

    def sample():
        while not stop.is_set():
            p, s = Hwmon.read(hw.power, 1e6), Hwmon.read(hw.sclk, 1e6)
            if p:
                pw.append(p)
            if s:
                sk.append(s)
            for label, path in hw.temps.items():
                t = Hwmon.read(path, 1e3)
                if t is not None:
                    hw.last_temps.setdefault(label, []).append(t)
            time.sleep(0.02)

    t = threading.Thread(target=sample)
    launch(); dev.sync()
    t.start()
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(8):
            launch()
        dev.sync(); n += 8
    dt = time.time() - t0
    stop.set(); t.join()
    med = lambda v: float(np.median(v)) if v else None  # noqa: E731
    return dt / n * 1e3, med(pw), med(sk)


_PINNED = {}


def pinned_u64(lib, words):
    """A pinned host array for the stamp read-back (the ABI's own allocator): the copy is a plain DMA into pages the
    driver already maps, whatever the runtime would do for pageable memory of this size."""
    if words not in _PINNED:
        p = C.c_void_p()
        assert lib.vkmr_hip_host_alloc(C.c_size_t(8 * words), C.byref(p)) == 0
        _PINNED[words] = (p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(words,)))
    return _PINNED[words]


def read_stamps(L, lib, tag, raw=None):
    if raw is None:
        p, view = pinned_u64(lib, SLOTS * 8)
        assert L.vkmr_hip_debug_stamps(p, SLOTS * 8) == 0      # (the read clears the buffer: a second kernel's stamps come from the same copy, `raw`)
        raw = view.copy().reshape(SLOTS, 8)
    s = raw
    s = s[((s[:, 6] & np.uint64(0xFFFFFF)) == tag) & (s[:, 5] > s[:, 1])]
    if len(s):
        s = s[s[:, 7] == s[:, 7].max()]      # the widest launch of that kernel (bulk pass 0), not the later, smaller ones
    return s.astype(np.float64)


def clock_of(s):
    ghz = (s[:, 4] - s[:, 0]) / (s[:, 5] - s[:, 1]) * 0.1
    return {"GHz_median": round(float(np.median(ghz)), 4), "GHz_p5": round(float(np.percentile(ghz, 5)), 4),
            "GHz_p95": round(float(np.percentile(ghz, 95)), 4), "workgroups_stamped": int(len(s)),
            "cycles_per_workgroup_median": float(np.median(s[:, 4] - s[:, 0]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves-log2", type=int, default=24)
    ap.add_argument("--maxlen", type=int, default=127)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--settle", type=int, default=12, help="launches between the stamp buffer's clearing and the launch whose stamps are read")
    ap.add_argument("--alternate", action="store_true",
                    help="one sustained loop of map, reduce, map, reduce ... (the bench's step) instead of one loop per kernel: the clock each kernel "
                         "holds when it runs behind the other one")
    ap.add_argument("--cus-per-engine", type=int, default=0,
                    help="1..7: run on a stream whose CU mask leaves that many CUs per shader engine (of 8; mask bit i = XCC i %% 8, engine (i / 8) %% 4, "
                         "alive CU (i / 8) / 4: profiles/r04_cu_mask_probe.txt) -- the kernels' cycles per instruction with the board far from its power cap")
    a = ap.parse_args()
    dev = vk.HipDevice(0)
    L = C.CDLL(STAMPS_LIB)
    if a.cus_per_engine:
        words = (C.c_uint32 * 8)(*[sum(1 << k for k in range(32) if 32 * w + k < 32 * a.cus_per_engine) for w in range(8)])
        masked = C.c_void_p()
        rc = C.CDLL("libamdhip64.so").hipExtStreamCreateWithCUMask(C.byref(masked), 8, words)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
        dev.stream = masked.value          # every launch, event and wait of this tool goes to the device's default stream
    pci = None
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            pci = buf.value.decode()
    except OSError:
        pass
    hw = Hwmon(pci)
    b = vk.rndm_packed(a.seed, 1 << a.leaves_log2, a.maxlen)
    d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * b.count)
    d_scr, d_root = dev.reduce_scratch(b.count), dev.alloc(32)
    height = vk.tree_height(b.count)
    # every device range this tool's kernels may touch, on stderr BEFORE the first launch: a memory access fault names an
    # address and nothing else (profiles/r04_kernel_clock_fault.txt)
    for name, buf in (("data", d_data), ("meta", d_meta), ("digests", d_out), ("reduce scratch", d_scr), ("root", d_root)):
        sys.stderr.write("[kernel_clock] %-14s 0x%012x .. 0x%012x (%d bytes)\n" % (name, buf.ptr, buf.ptr + buf.nbytes, buf.nbytes))
    sp, _ = pinned_u64(dev.lib, SLOTS * 8)
    sys.stderr.write("[kernel_clock] %-14s 0x%012x .. 0x%012x (pinned host)\n" % ("stamp copy", sp.value, sp.value + SLOTS * 64))
    sys.stderr.flush()
    out = {"cus": 32 * a.cus_per_engine if a.cus_per_engine else 256, "workload": f"rndm {a.seed} 2^{a.leaves_log2} {a.maxlen}", "library": os.path.relpath(STAMPS_LIB, ROOT),
           "method": "per-workgroup d(s_memtime)/d(s_memrealtime) x 100 MHz after >= %.1f s of back-to-back launches" % a.seconds,
           "power_cap_W": Hwmon.read(hw.cap, 1e6), "hwmon": hw.base}
    e0, e1 = dev.new_event(), dev.new_event()

    def one(name, tag, launch):
        ms_sus, pw, sk = sustained(dev, launch, a.seconds, hw)
        L.vkmr_hip_debug_stamps(None, 0)   # clear (a device-wide wait on either side: the chip idles for a moment)
        # The stamps that are read are those of the LAST of several back-to-back launches (each overwrites the one before): the
        # first launch after the pause runs on a clock that is still ramping (2^26: 6.5 ms against 5.3 sustained, 1.69 GHz
        # against 2.0 -- profiles/r04_kernel_clock_fault.txt)
        for _ in range(a.settle):
            launch()
        dev.record(e0); launch(); dev.record(e1); dev.sync()
        s = read_stamps(L, dev.lib, tag)
        rec = clock_of(s)
        if hw.last_temps:
            rec["temperature_C_last_half_second"] = {k: round(float(np.median(v[-25:])), 1) for k, v in hw.last_temps.items()}
        rec.update(ms_per_launch_sustained=round(ms_sus, 4), ms_stamped_launch=round(dev.elapsed_ms(e0, e1), 4),
                   board_power_W=pw, sclk_sysfs_MHz=sk, kernel=dev.lib.vkmr_hip_kernel_info().decode().split(" reduce=")[0] if tag == 0x4d4150 else "reduce_pass_kernel (first bulk pass)")
        if tag == 0x4d4150 and len(s):   # phases of a map workgroup's first wavefront
            life = s[:, 4] - s[:, 0]
            wait = (s[:, 6].astype(np.uint64) >> np.uint64(24)).astype(np.float64)   # persistent form: cycles at the tile-done barrier
            if wait.max() > 0:
                rec["barrier_wait_share"] = round(float(np.median(wait / life)), 4)
            rec["phase_share"] = {"sort": round(float(np.median((s[:, 2] - s[:, 0]) / life)), 4),
                                  "stage": round(float(np.median((s[:, 3] - s[:, 2]) / life)), 4),
                                  "hash": round(float(np.median((s[:, 4] - s[:, 3]) / life)), 4)}
        out[name] = rec

    def do_map():
        dev.map_async(d_data, b.words, d_meta, b.count, d_out)

    def do_reduce():
        dev.reduce_async(d_out, b.count, height, d_scr, d_root)

    if a.alternate:
        def step():
            do_map(); do_reduce()
        ms_sus, pw, sk = sustained(dev, step, a.seconds, hw)
        L.vkmr_hip_debug_stamps(None, 0)
        for _ in range(a.settle):
            step()
        e2 = dev.new_event()
        dev.record(e0); do_map(); dev.record(e1); do_reduce(); dev.record(e2); dev.sync()
        out["alternating"] = {"ms_per_step_sustained": round(ms_sus, 4), "board_power_W": pw, "sclk_sysfs_MHz": sk,
                              "temperature_C_last_half_second": {k: round(float(np.median(v[-25:])), 1) for k, v in hw.last_temps.items()},
                              "temperature_C_first_samples": {k: round(float(np.median(v[:3])), 1) for k, v in hw.last_temps.items()},
                              "map_ms_stamped_step": round(dev.elapsed_ms(e0, e1), 4), "reduce_ms_stamped_step": round(dev.elapsed_ms(e1, e2), 4)}
        p, view = pinned_u64(dev.lib, SLOTS * 8)
        assert L.vkmr_hip_debug_stamps(p, SLOTS * 8) == 0
        raw = view.copy().reshape(SLOTS, 8)
        for name, tag in (("map_kernel", 0x4d4150), ("reduce_pass_kernel", 0x524544)):
            out[name] = clock_of(read_stamps(L, dev.lib, tag, raw))
        out["map_kernel"]["kernel"] = dev.lib.vkmr_hip_kernel_info().decode().split(" reduce=")[0]
        print(json.dumps(out))
        return
    one("map_kernel", 0x4d4150, do_map)
    one("reduce_pass_kernel", 0x524544, do_reduce)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
