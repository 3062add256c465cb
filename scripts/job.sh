mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
tail -c 3600 gpurun_out/bench_default.log; tail -5 gpurun_out/bench_default.err
