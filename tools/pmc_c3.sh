# Counter passes over the textured 1080p / 64 spp workload (config 3): SQ instruction and wait counters, L1 / L2 hit
# counters, per kernel.  One rocprofv3 run per pass; prints per-kernel sums and writes gpurun_out/pmc_c3/*.txt.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/pmc_c3
rm -rf $out && mkdir -p $out
run() { name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --timed-frames-only --workload textured_1080p_64spp --steps 1 --warmup 0 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
  python3 tools/pmc_sum.py $out/$name/p_counter_collection.csv > $out/$name.txt || exit 1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT && \
run cache TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE && \
cat $out/sq1.txt $out/sq2.txt $out/cache.txt
