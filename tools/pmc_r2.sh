# Counter passes over the reduced path-traced workload (tools/prof_c4.py: 1M triangles, 960x540, 16 spp, wide tree):
# kernel statistics, then L2 / L1 counters, SQ instruction and busy counters, and the L1 address / data path.
# Prints per-kernel sums; writes gpurun_out/pmc_r2_*.txt.  One rocprofv3 run per pass.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/pmc_r2
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c4 -- python3 tools/prof_c4.py > $out/stats.log 2>&1 || { echo stats failed; tail -3 $out/stats.log; exit 1; }
rm -f $out/stats/c4_kernel_trace.csv
cut -c1-160 $out/stats/c4_kernel_stats.csv > $out/kernel_stats.txt
run() { name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/prof_c4.py > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
  python3 tools/pmc_sum.py $out/$name/p_counter_collection.csv > $out/$name.txt || exit 1
}
run cache TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum && \
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT && \
run l1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_BUSY_avr GRBM_GUI_ACTIVE && \
cat $out/kernel_stats.txt $out/cache.txt $out/sq1.txt $out/sq2.txt $out/l1.txt
