# HBM traffic of the sweep kernels at the level-0 size of the 2000x4000 strip (single stream; bounded by timeouts)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcs_$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcs_$c -o p -- python tests/micro/gpu_sweep_bench.py 1100x2000 > gpurun_out/pmcs_$c.log 2>&1
  echo "$c rc=$?"
done
python - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('gpurun_out/pmcs_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'sweep' in r['Kernel_Name']:
                print(c, r['Kernel_Name'][:24], "grid", r['Grid_Size'], "value_KB", r['Counter_Value'])
PY
