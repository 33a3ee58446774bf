cd $GRAFT_REPO_ROOT
for ml in 200 300 600 1200; do
  for v in 0 4 1; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py 22 $ml; done
done
