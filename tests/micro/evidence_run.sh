#!/bin/bash
# evidence run of a round (default r06): full bench line, driver-command line, rocprofv3 kernel stats + FETCH / WRITE PMC of the bench command, SQ counters of
# the lone pair's sweeps, batch profile + PMC + dynamic instruction counts; everything lands in gpurun_out/<R>_* (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err
grep '^{"metric"' gpurun_out/bench_full.log | tail -1 > gpurun_out/${R}_bench_full_line.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench_driver.log 2> gpurun_out/bench_driver.err
grep '^{"metric"' gpurun_out/bench_driver.log | tail -1 > gpurun_out/${R}_bench_driver_cmd_line.json
bash tests/micro/refresh_profiles.sh $R > gpurun_out/refresh_$R.log 2>&1
bash tests/micro/sweep_sq.sh $R > gpurun_out/sweep_sq_$R.log 2>&1
bash tests/micro/batch_profile.sh $R 9000 4000 1 > gpurun_out/batch_profile_$R.log 2>&1
bash tests/micro/batch_insts.sh > gpurun_out/${R}_batch_instruction_counts.txt 2> gpurun_out/batch_insts_$R.err
cut -c1-1500 gpurun_out/${R}_bench_full_line.json
cut -c1-300 gpurun_out/${R}_bench_driver_cmd_line.json
tail -5 gpurun_out/refresh_$R.log | cut -c1-300
tail -12 gpurun_out/sweep_sq_$R.log | cut -c1-200
tail -25 gpurun_out/batch_profile_$R.log | cut -c1-200
tail -22 gpurun_out/${R}_batch_instruction_counts.txt | cut -c1-200
tail -3 gpurun_out/bench_full.err
