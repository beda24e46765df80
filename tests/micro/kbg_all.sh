for k in k_gauss15 k_median5 k_flow_grad k_sweep_prep k_sweep_t k_warp k_bgra k_pyr; do bash tests/micro/kern_by_grid.sh $k 6 > gpurun_out/kbg_$k.txt 2>&1; done
cat gpurun_out/kbg_k_*.txt
