cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o c4 -- python3 tools/prof_c4.py > gpurun_out/prof_c4.log 2>&1
find gpurun_out/prof_c4 -name "*kernel_stats*" -exec cat {} \;
