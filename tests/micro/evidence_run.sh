#!/bin/bash
# round-5 evidence run: full bench line, driver-command line, rocprofv3 kernel stats + PMC of the bench command, batch profile + PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err
grep '^{"metric"' gpurun_out/bench_full.log | tail -1 > gpurun_out/r05_bench_full_line.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench_driver.log 2> gpurun_out/bench_driver.err
grep '^{"metric"' gpurun_out/bench_driver.log | tail -1 > gpurun_out/r05_bench_driver_cmd_line.json
bash tests/micro/refresh_profiles.sh r05 > gpurun_out/refresh_r05.log 2>&1
bash tests/micro/batch_profile.sh r05 9000 4000 1 > gpurun_out/batch_profile_r05.log 2>&1
cut -c1-1500 gpurun_out/r05_bench_full_line.json
cut -c1-300 gpurun_out/r05_bench_driver_cmd_line.json
tail -5 gpurun_out/refresh_r05.log | cut -c1-300
tail -25 gpurun_out/batch_profile_r05.log | cut -c1-200
tail -3 gpurun_out/bench_full.err
