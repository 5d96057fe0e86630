// continuity_lds.h -- host interface of the LDS-resident mass-flux kernel (continuity_lds.hip).
#pragma once
#include "continuity_dev.h"

struct LdsArgs {
  double h_min;        // 2*Angstrom_H, the positive-definiteness floor of PPM_limit_pos
  int scheme;          // 0 PPM, 1 simple_2nd, 2 upwind_1st
  int monotonic;       // PPM_limit_CW84 instead of PPM_limit_pos
  int marginal;        // BT_cont%h_u from the marginal (not the average) face thickness
  int fma;             // continuity_wave.hip: sum_order == MOM6X_SUM_TREE16_FMA (fused multiply-adds at fixed sites)
  double *h_face;      // BT_cont%h_u | h_v (3-D) or null
  int gx, gy, rows;    // tile grid and tile rows per XCD band (set by mass_flux_lds)
  int i_base;          // first i of the tile grid (set by mass_flux_lds: 128-byte aligned, <= a0)
  int force_walk;      // tests (MOM6X_MASSFLUX=lds_walk): take the sequential duL/duR recurrence even when the certificate holds
  int *retry;          // per tile: the cheap-bounds pass asks for the exact pass (set by mass_flux_lds), or null
  // continuity_wave.hip: the face ranges of the launch, one or two parts (the two rims of a split pass go out as ONE launch); a part
  // is pgx x pgy work-groups of 16 faces x `rows` rows starting at the 128-byte aligned pib
  int np, pa0[2], pa1[2], pb0[2], pb1[2], pib[2], pgx[2], pgy[2];
  int no_pairs;        // continuity_wave.hip: the results are not 16-byte aligned (odd slab or a host array at an odd double): 8-byte stores
  unsigned long long *stats;   // continuity_wave.hip: [0] flux re-evaluations of all Newton solves, [1] solves (face columns x solves), [2] exact-limit redos
};

size_t mass_flux_lds_bytes(int dir, int nk);
bool mass_flux_lds_usable(int nk);
int mass_flux_lds(mom6x_ctx *c, int dir, const FluxArgs &A, const LdsArgs &E);
// continuity_wave.hip: the wave-owned kernel of sum_order == MOM6X_SUM_TREE16 (uses h_min, scheme, monotonic, marginal, h_face)
bool mass_flux_wave_usable(int nk);
int mass_flux_wave(mom6x_ctx *c, int dir, const FluxArgs &A, const LdsArgs &E);
// ... over two face ranges in one launch (A2's ranges; everything else from A); an empty range is left out
int mass_flux_wave_pair(mom6x_ctx *c, int dir, const FluxArgs &A, const FluxArgs &A2, const LdsArgs &E);
