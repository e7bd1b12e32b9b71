# Runs the default benchmark and a rocprofv3 kernel trace of a shorter run of the same command.
# usage (on the GPU box): bash tools/bench_profile.sh <tag>
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
tag=${1:-r1}
timeout -k 10 900 python -u bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo bench failed; tail -5 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || { echo rocprof failed; tail -5 gpurun_out/prof_$tag.log; exit 1; }
rm -f gpurun_out/prof_$tag/bench_kernel_trace.csv
cat gpurun_out/prof_$tag/bench_kernel_stats.csv | cut -c1-200
