cd $GRAFT_REPO_ROOT
for c in 2097152 4194304 8388608 16777216; do
  echo "capacity $c: $(RT_WAVEFRONT_CAPACITY=$c timeout -k 10 120 python3 tools/prof_c4.py | tail -1)" || exit 1
done
