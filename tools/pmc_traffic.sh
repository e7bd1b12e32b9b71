# HBM-side traffic of every kernel of the reduced path-traced workload: FETCH_SIZE and WRITE_SIZE in separate passes.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 tools/prof_c4.py > gpurun_out/pmc_fetch.log 2>&1 || { echo fetch pass failed; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 tools/prof_c4.py > gpurun_out/pmc_write.log 2>&1 || { echo write pass failed; exit 1; }
python3 tools/pmc_sum.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv
