# A/B of environment settings on the GPU box, textured 1080p / 64 spp frame (config 3): tools/ab_env_c3.sh "" "RT_TRACE_LEAF_MIN=8" ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== ${e:-default}"
  env $e timeout -k 10 200 python3 -u bench.py --timed-frames-only --workload textured_1080p_64spp --steps 2 --warmup 1 || exit 1
done
