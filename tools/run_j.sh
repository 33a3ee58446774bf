cd $GRAFT_REPO_ROOT
python3 tools/long_strings_cached_probe.py
python3 tools/long_strings_cached_probe.py
