cd $GRAFT_REPO_ROOT
AB_TOUCH=mom6_amd/csrc/hor_visc.hip AB_STEPS=4 bash scripts/dev/ab_kernels.sh "k_hv_fused" "" "-DHV_TY=12 -DHV_WAVES=3" "-DHV_TX=16 -DHV_TY=24 -DHV_WAVES=3" "-DHV_TY=8 -DHV_WAVES=3" "-DHV_TX=64 -DHV_TY=12" "" > gpurun_out/r06_ab_hv.txt 2>&1
