cd $GRAFT_REPO_ROOT
for c in 16777216 33554432 67108864; do
  echo "capacity $c: $(RT_WAVEFRONT_CAPACITY=$c timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline | cut -c1-220)" || exit 1
done
