cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2; do
echo "# cat file | vkmr hip:0"
cat /tmp/g25.txt | VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 2>&1 | grep -E "computed|read \(|pass 1|pass 2"
echo "# dd bs=1M | vkmr (pipe 1024)"
dd if=/tmp/g25.txt bs=1M 2>/dev/null | VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 2>&1 | grep -E "computed|read \("
echo "# rndm | vkmr"
vk_merkle_roots_amd/bin/rndm 42 33554432 127 2>/dev/null | VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 2>&1 | grep -E "computed|read \("
done
echo "# the writers alone: cat > /dev/null through a pipe read by python with 16 MiB reads; rndm > /dev/null"
python3 -c "
import subprocess, time
for cmd in (['cat','/tmp/g25.txt'], ['dd','if=/tmp/g25.txt','bs=1M'], ['vk_merkle_roots_amd/bin/rndm','42','33554432','127']):
    t=time.time(); p=subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL); n=0
    while True:
        b=p.stdout.read(1<<24)
        if not b: break
        n+=len(b)
    print(cmd[0], '%d bytes through the pipe in %.3f s' % (n, time.time()-t))"
} > gpurun_out/r03/pipe.txt 2>&1
cat gpurun_out/r03/pipe.txt
