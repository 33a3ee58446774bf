cd $GRAFT_REPO_ROOT
for ml in 300 400 540 1200; do for v in 0 4 7; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py 22 $ml; done; done
