cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 ./tools/issue_patterns 1.0 > gpurun_out/r03/issue_patterns_set5_power.txt 2>&1; echo "issue_patterns rc=$?"
