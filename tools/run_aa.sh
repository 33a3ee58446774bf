cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --json | head -c 600; echo
python3 tools/clock_power_probe.py
