# One kernel-stats pass and two SQ counter passes over tools/prof_c4.py; prints per-kernel sums.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
rm -rf gpurun_out/prof_c4 gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o c4 -- python3 tools/prof_c4.py > gpurun_out/prof_c4.log 2>&1 || exit 1
rm -f gpurun_out/prof_c4/c4_kernel_trace.csv
cut -c1-150 gpurun_out/prof_c4/c4_kernel_stats.csv
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o p -- python3 tools/prof_c4.py > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; exit 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT && \
python3 tools/pmc_sum.py gpurun_out/pmc_sq1/p_counter_collection.csv gpurun_out/pmc_sq2/p_counter_collection.csv
