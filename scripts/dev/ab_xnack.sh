#!/bin/bash
# dev A/B: the library compiled for gfx950:xnack- next to the generic gfx950 code object (the loader takes the more specific one)
cd $GRAFT_REPO_ROOT
AB_STEPS=6 bash scripts/dev/ab_kernels.sh "k_mass_flux_wave<0|k_corad|k_bt_col<0, true|k_vertvisc_coef_cols<0, 3, true, true|k_hv_fused|k_pgf_main|k_bt_vel<0|k_convergence<0" "ARCH=gfx950:xnack-" "" > gpurun_out/r06_ab_xnack.txt 2>&1
