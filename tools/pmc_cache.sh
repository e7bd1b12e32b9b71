# Cache counters of every kernel of the reduced path-traced workload: L2 hits / misses and L1 accesses / L1-to-L2 reads.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
rm -rf gpurun_out/pmc_cache
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmc_cache -o p -- python3 tools/prof_c4.py > gpurun_out/pmc_cache.log 2>&1 || { echo cache pass failed; tail -5 gpurun_out/pmc_cache.log; exit 1; }
python3 tools/pmc_sum.py gpurun_out/pmc_cache/p_counter_collection.csv
