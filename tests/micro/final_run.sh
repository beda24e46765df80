cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/gpu_tests_d.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_d.log 2>&1
timeout 800 python bench.py > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err
rm -rf gpurun_out/prof_r03
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o bench -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_r03.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_r03.log | tail -1 > gpurun_out/r03_bench_line.json
cp $(find gpurun_out/prof_r03 -name '*kernel_stats.csv' | head -1) gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r03
cat gpurun_out/gpu_tests_d.log; tail -2 gpurun_out/smoke_d.log
