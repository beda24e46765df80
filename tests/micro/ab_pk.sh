#!/bin/bash
# same-box A/B: the sweep's packed chains as asm blocks (product) vs compiler-scheduled (-DPF_SAFE_PK); var_libs/lib_ab_asm.so / lib_ab_safepk.so
cd $GRAFT_REPO_ROOT
bash tests/micro/ab3.sh 3 2>&1 | cut -c1-150
