cd $GRAFT_REPO_ROOT
( time bash tools/refresh_profiles.sh r03 ) > gpurun_out/refresh_r03.log 2>&1
tail -4 gpurun_out/refresh_r03.log
