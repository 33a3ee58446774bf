cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/rndm 42 67108864 127 > /tmp/g26.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
python3 - <<'PY'
import subprocess, time, statistics
for name, path, n in (("2^25 strings (2.13 GB)", "/tmp/g25.txt", 24), ("2^26 strings (4.26 GB)", "/tmp/g26.txt", 12)):
    printed, wall = [], []
    for _ in range(n):
        t = time.time()
        r = subprocess.run(["vk_merkle_roots_amd/bin/vkmr", "hip:0"], stdin=open(path, "rb"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        wall.append(time.time() - t)
        line = [l for l in r.stdout.decode().splitlines() if "computed root" in l][-1]
        printed.append(float(line.rsplit(" in ", 1)[1]))
    q = statistics.quantiles(printed, n=4)
    print(f"vkmr hip:0 < {name}, {n} runs: printed min {min(printed):.1f}  quartiles {q[0]:.1f} / {q[1]:.1f} / {q[2]:.1f}  max {max(printed):.1f} ms; process wall median {statistics.median(wall):.3f} s (min {min(wall):.3f})")
    print("  printed:", " ".join(f"{p:.1f}" for p in printed))
PY
} > gpurun_out/r03/e2e_distribution.txt 2>&1
cat gpurun_out/r03/e2e_distribution.txt
