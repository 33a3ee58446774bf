"""Shader clock and board power while each kernel runs back to back for a few seconds (rocm-smi sampled from a side
thread): does the long-string map kernel run at a lower clock than the same strings L2-resident, as DESIGN.md says?
GPU box only:  python3 tools/clock_power_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_merkle_roots_amd as vk  # noqa: E402

dev = vk.HipDevice(0)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=5)
            card = next(iter(json.loads(r.stdout.decode()).values()))
            sclk = [v for k, v in card.items() if "sclk" in k.lower()]
            pwr = [v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()]
            out.append((sclk[0] if sclk else None, pwr[0] if pwr else None))
        except Exception as e:   # keep sampling
            out.append((None, repr(e)[:60]))
        time.sleep(0.05)


def run(name, launch, seconds=5.0):
    stop, out = threading.Event(), []
    t = threading.Thread(target=sample, args=(stop, out))
    launch(); dev.sync()
    t.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(8):
            launch()
        dev.sync(); n += 8
    dt = time.time() - t0
    stop.set(); t.join()
    clk = [float(str(c).strip("()Mhz ")) for c, _ in out if c not in (None, "")]
    pw = [float(p) for _, p in out if p not in (None, "") and str(p).replace(".", "", 1).isdigit()]
    print(f"{name}: {dt / n * 1e3:.3f} ms per launch; sclk median {np.median(clk) if clk else float('nan'):.0f} MHz (min {min(clk) if clk else 0:.0f}, max {max(clk) if clk else 0:.0f}), "
          f"power median {np.median(pw) if pw else float('nan'):.0f} W over {len(out)} samples; first raw sample {out[0] if out else None}")


b = vk.rndm_packed(42, 1 << 23, 127)
d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * b.count)
run("map, rndm 42 2^23 127 (LDS-staged tiles)", lambda: dev.map_async(d_data, b.words, d_meta, b.count, d_out))
d_scr, d_root = dev.reduce_scratch(b.count), dev.alloc(32)
run("reduce, one slice of 2^23 digests", lambda: dev.reduce_async(d_out, b.count, 23, d_scr, d_root))
for x in (d_data, d_meta, d_out):
    x.free()
L = vk.rndm_packed(42, 1 << 21, 4096)
d_data, d_out = dev.upload(L.data), dev.alloc(32 * L.count)
d_meta = dev.upload(L.meta)
run("map, rndm 42 2^21 4096 streamed (line window)", lambda: dev.map_async(d_data, L.words, d_meta, L.count, d_out))
folded = L.meta.copy(); folded[:, 0] = folded[:, 0] % np.uint32(1 << 18)
d_meta2 = dev.upload(folded)
run("map, the same strings with starts folded into 1 MiB (L2-resident)", lambda: dev.map_async(d_data, L.words, d_meta2, L.count, d_out))
