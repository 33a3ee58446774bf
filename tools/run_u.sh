cd $GRAFT_REPO_ROOT
python - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import build_virt_devices
os.environ.update(LD_PRELOAD=build_virt_devices(), VKMR_TEST_VIRTUAL_DEVICES="8")
import bench
print(bench.hip_all_check(60))
PY
