# FETCH_SIZE per kernel over tools/frame_4k64.py, once per environment setting given: tools/pmc_fetch_4k64.sh "" "RT_SLOT_ORDER=0"
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for e in "$@"; do
  echo "== ${e:-default}"
  rm -rf gpurun_out/pmc_f4k
  ( [ -n "$e" ] && export $e; timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f4k -o p -- python3 tools/frame_4k64.py > gpurun_out/pmc_f4k.log 2>&1 ) || { echo pass failed; tail -3 gpurun_out/pmc_f4k.log; exit 1; }
  python3 tools/pmc_sum.py gpurun_out/pmc_f4k/p_counter_collection.csv | grep -A1 "trace_kernel\|shade_kernel"
  rm -f gpurun_out/pmc_f4k/p_counter_collection.csv
done
