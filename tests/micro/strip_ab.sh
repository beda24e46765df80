#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_<name>.so: lone 2000x4000 strip and the x8-displacement dense pair: strip_ab.sh names...
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in "$@"; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo -n "== $v: "
DISP_COLS=2000 DISP_ROWS=4000 DISP_INFLIGHT=0 timeout 600 python tests/micro/disp_probe.py 1 2>&1 | grep "lone" | sed 's/.*lone pair: \([0-9.]*\) ms.*/strip \1 ms/' | tr '\n' ' '
DISP_INFLIGHT=0 timeout 600 python tests/micro/disp_probe.py 8 2>&1 | grep "lone" | sed 's/.*lone pair: \([0-9.]*\) ms.*/pair x8 \1 ms/'
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
