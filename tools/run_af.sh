cd $GRAFT_REPO_ROOT
for ml in 127 160 200 250; do for v in 0 4; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py 23 $ml; done; done
