cd $GRAFT_REPO_ROOT
for ml in 200; do for f in 0 75 83 93; do VKMR_MAP_FIT=$f python3 tools/long_strings_probe.py 23 $ml; done; done
for ml in 160; do for f in 0 60 75 89 96; do VKMR_MAP_FIT=$f python3 tools/long_strings_probe.py 23 $ml; done; done
