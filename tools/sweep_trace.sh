cd $GRAFT_REPO_ROOT
for r in 8 16 24 32; do for l in 8 16 32; do
  echo "refill $r leaf $l: $(RT_TRACE_REFILL_MIN=$r RT_TRACE_LEAF_MIN=$l timeout -k 10 120 python3 tools/prof_c4.py | tail -1)" || exit 1
done; done
