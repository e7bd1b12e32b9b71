# Runs the default benchmark, then a rocprofv3 kernel trace of the same timed frames (bench.py --timed-frames-only:
# without the CPU leg, the counted frame and the frame without ray sharing, whose launches would mix into the averages).
# usage (on the GPU box): bash tools/bench_profile.sh <tag>
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
tag=${1:-r1}
timeout -k 10 900 python -u bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo bench failed; tail -5 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
rm -rf gpurun_out/prof_$tag
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --timed-frames-only > gpurun_out/prof_$tag.log 2>&1 || { echo rocprof failed; tail -5 gpurun_out/prof_$tag.log; exit 1; }
rm -f gpurun_out/prof_$tag/bench_kernel_trace.csv
cut -c1-200 gpurun_out/prof_$tag/bench_kernel_stats.csv
tail -1 gpurun_out/prof_$tag.log | cut -c1-1500
