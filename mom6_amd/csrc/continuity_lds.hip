// continuity_lds.hip -- the mass-flux half of continuity_PPM as ONE LDS-resident kernel per direction.
//
// Replaces, for a tile of NF face columns, the sequence k_edge -> k_mass_flux -> k_flux_thickness of
// continuity.hip (PPM_reconstruction_x/y :2307/:2442, zonal/meridional_mass_flux :519/:1412 with
// zonal/meridional_flux_adjust :1093/:1992 and set_zonal/merid_BT_cont :1246/:2143,
// zonal/merid_flux_thickness :975/:1866 of MOM_continuity_PPM.F90).
//
// Why: the Newton iteration of flux_adjust re-reads every layer of a face column up to 20 times (and
// set_*_BT_cont sweeps it another ~8 times).  The thread-per-column kernel did those re-reads through
// L2/HBM.  Here a work-group of NF(16 faces along i) x KL(16 layer lanes) threads
//   * reconstructs the PPM edge values of the cells it needs straight from h (all global loads of a lane
//     are issued back to back) and parks (h_L, h_R, curvature) of all nk layers in LDS
//     (<= 77 KB at nk = 75; 160 KB per CU on gfx950),
//   * keeps u and visc_rem of its own layers in registers,
//   * evaluates the layer fluxes with all 256 lanes, and lets 16 "face lanes" of ONE wavefront do the
//     column sums and recurrences SEQUENTIALLY in k -- the reference's order, so the result is
//     bit-identical to the Fortran loop nest (and to the thread-per-column kernel).  Everything that
//     does not carry a k-dependence (the divisions of the CFL / duL / duR recurrences, visc_rem_max)
//     is computed by all lanes first, so the sequential walks are a few instructions per layer,
//   * runs the Newton loop work-group-uniformly: a face whose do_I is false is frozen exactly as in the
//     reference's row-wide loop,
//   * writes uh, u_cor, du_cor, the BT_cont fits and h_u/h_v once.
// HBM traffic per face-layer: read h (+halo re-reads from L2), u, visc_rem; write uh, u_cor, h_u.
#include "continuity_dev.h"
#include "continuity_lds.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int NF = 16;        // faces along i per work-group (the coalesced axis)
#ifndef MOM6X_X_WG
#define MOM6X_X_WG 2   // work-groups per CU the zonal kernel is compiled for: LDS allows 3 at nk = 75, but 168 VGPRs spill (same speed, +1 GB of scratch traffic per launch)
#endif

// Dev tool (MOM6X_CFLAGS=-DMOM6X_MFL_TIMING python -m mom6_amd.build --force; scripts/prof_continuity.py): shader-clock
// cycles lane 0 of every work-group spends in each phase of the kernel, summed over the work-groups.
#ifdef MOM6X_MFL_TIMING
__device__ unsigned long long g_mfl_t[2][16];
#define TICK_INIT long long t_prev_ = clock64()
#define TICK(p) do { if (threadIdx.x == 0) { const long long t_ = clock64(); atomicAdd(&g_mfl_t[DIR][p], (unsigned long long)(t_ - t_prev_)); t_prev_ = t_; } } while (0)
#else
#define TICK_INIT
#define TICK(p)
#endif
// KL = layer lanes per face (template parameter): 16, i.e. 4 wavefronts per work-group

// zonal_flux_layer :896 / merid_flux_layer :1787 with the cell data in LDS.
// m / p: LDS slots of the minus / plus cell of the face.
__device__ __forceinline__ void flux_lds(double u, double dt, double IdT_m, double IdT_p, double vrem, double Lf,
                                         const double *sL, const double *sR, const double *sC, int m, int p,
                                         double &uh, double &duhdu) {
  // The two upwind branches of the reference are mirror images: with a = the edge value on the face side
  // of the upwind cell and b = the far one, both read  a + CFL*(0.5*(b-a) + curv_3*(CFL-1.5)).  Selecting
  // the operands instead of branching keeps a wavefront with mixed flow directions from running both.
  double h_marg;
  if (u != 0.0) {
    const bool pos = (u > 0.0);
    const int c = pos ? m : p;
    const double l = sL[c], r = sR[c], curv_3 = sC[c];
    const double a = pos ? r : l, b = pos ? l : r;
    const double CFL = (pos ? u : -u) * dt * (pos ? IdT_m : IdT_p);
    uh = Lf * u * (a + CFL * (0.5 * (b - a) + curv_3 * (CFL - 1.5)));
    h_marg = a + CFL * ((b - a) + 3.0 * curv_3 * (CFL - 1.0));
  } else {
    uh = 0.0;
    h_marg = 0.5 * (sL[p] + sR[m]);
  }
  duhdu = Lf * h_marg * vrem;
}

// ---- sequential walks of one face column through the LDS transit arrays -----------------------------
// The loads of a batch are independent and issued together; the caller's recurrence then runs on
// registers, so the LDS latency is paid once per batch instead of once per layer.
template <int U, typename F>
__device__ __forceinline__ void batch2(const double *a, const double *b, int k0, F &f) {
  double x[U], y[U];
#pragma unroll
  for (int q = 0; q < U; q++) { x[q] = a[(k0 + q) * NF]; y[q] = b[(k0 + q) * NF]; }
#pragma unroll
  for (int q = 0; q < U; q++) f(x[q], y[q]);
}
template <typename F>
__device__ __forceinline__ void col_walk2(const double *a, const double *b, int n, F f) {   // a, b already offset by fl
  int k = 0;
  for (; k + 16 <= n; k += 16) batch2<16>(a, b, k, f);
  for (; k + 4 <= n; k += 4) batch2<4>(a, b, k, f);
  for (; k < n; k++) batch2<1>(a, b, k, f);
}
template <int U, typename F>
__device__ __forceinline__ void batch4(const double *a, int str, int k0, F &f) {
  double x[U], y[U], z[U], w[U];
#pragma unroll
  for (int q = 0; q < U; q++) {
    const double *p = a + (k0 + q) * NF;
    x[q] = p[0]; y[q] = p[str]; z[q] = p[2 * str]; w[q] = p[3 * str];
  }
#pragma unroll
  for (int q = 0; q < U; q++) f(x[q], y[q], z[q], w[q]);
}
template <typename F>
__device__ __forceinline__ void col_walk4(const double *a, int str, int n, F f) {   // 4 arrays, `str` apart
  int k = 0;
  for (; k + 8 <= n; k += 8) batch4<8>(a, str, k, f);
  for (; k + 2 <= n; k += 2) batch4<2>(a, str, k, f);
  for (; k < n; k++) batch4<1>(a, str, k, f);
}

template <int U, typename F>
__device__ __forceinline__ void batch5(const double *a, const double *b, const double *c, int str, int k0, F &f) {
  double x[U], y[U], z[U], w[U], v[U];
#pragma unroll
  for (int q = 0; q < U; q++) {
    const int o = (k0 + q) * NF;
    x[q] = a[o]; y[q] = b[o]; z[q] = c[o]; w[q] = c[o + str]; v[q] = c[o + 2 * str];
  }
#pragma unroll
  for (int q = 0; q < U; q++) f(x[q], y[q], z[q], w[q], v[q]);
}
template <typename F>
__device__ __forceinline__ void col_walk5(const double *a, const double *b, const double *c, int str, int n, F f) {
  int k = 0;
  for (; k + 8 <= n; k += 8) batch5<8>(a, b, c, str, k, f);
  for (; k + 2 <= n; k += 2) batch5<2>(a, b, c, str, k, f);
  for (; k < n; k++) batch5<1>(a, b, c, str, k, f);
}

struct Tile {   // what every lane knows about its face and the LDS arrays
  const double *sL, *sR, *sC;
  double *sA, *sB;
  double *s_du; int *s_doI;
  int fl, kl, nk, NC, cm, cp;   // cm/cp: cell slots (within one layer) of the minus / plus cell
  bool lead;                    // this lane is the face lane of column fl (whether or not the face is active)
  double dt, IdT_m, IdT_p, Lf;
};

// A k-recurrence whose per-layer operands (4 of them) are produced by all lanes and consumed in k order
// by the face lane.  The 4 x [nk] operands go through the sA|sB space in two halves of KH = ceil(nk/2).
// Every lane of the work-group must call this.
template <int KL, int MAXL, typename P, typename S>
__device__ __forceinline__ void recurrence4(const Tile &T, bool face, P produce, S step) {
  const int KH = (T.nk + 1) >> 1, str = KH * NF;
  double *t = T.sA;   // sA and sB are contiguous: 2 * nkp * NF >= 4 * KH * NF
  for (int c = 0; c < 2; c++) {
    const int k_lo = c * KH, k_hi = (k_lo + KH < T.nk) ? k_lo + KH : T.nk;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = T.kl + KL * n;
      if (k >= k_lo && k < k_hi) {
        double o[4];
        produce(n, o);
        double *p = t + (k - k_lo) * NF + T.fl;
        p[0] = o[0]; p[str] = o[1]; p[2 * str] = o[2]; p[3 * str] = o[3];
      }
    }
    __syncthreads();
    if (face) col_walk4(t + T.fl, str, k_hi - k_lo, step);
    __syncthreads();
  }
}

// zonal_flux_adjust :1093-1242 / meridional_flux_adjust :1992-2140, iterated work-group-uniformly.
// Every lane of the work-group must call this.  Face lanes (face == true) carry the Newton state; all
// lanes re-evaluate the layer fluxes of the faces that are still iterating.  With store == true the
// last evaluated transports stay in sA (the uh_3d argument of the reference).
template <int KL, int MAXL>
__device__ __forceinline__ double wg_flux_adjust(const Tile &T, bool face, const double (&u_r)[MAXL],
                                                 const double (&v_r)[MAXL], double IareaMin, double uhbt,
                                                 double uh_tot_0, double duhdu_tot_0, double du_max,
                                                 double du_min, double tol_eta_cs, double tol_vel,
                                                 int better_iter, bool store, bool lazy, bool &need_exact) {
  // `lazy`: du_max / du_min are not the CFL limits themselves but a lower / an upper bound of them (see the
  // caller).  The limits only matter in the tests "du >= du_max" / "du <= du_min"; as long as every such test is
  // decided by the bound alone the result is the one the exact limits give.  The first test that is not ends the
  // solve for the whole work-group with need_exact = true, and the caller repeats it with the exact limits.
  const int max_itts = 20;
  double du = 0.0;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0;
  double uh_err_best = fabs(uh_err);
  bool do_I = face;
  bool max_lazy = lazy, min_lazy = lazy, undecided = false;
  need_exact = false;
  for (int itt = 1; itt <= max_itts; itt++) {
    if (do_I) {
      double tol_eta;
      if (itt <= 1) tol_eta = 1e-6 * tol_eta_cs;
      else if (itt == 2) tol_eta = 1e-4 * tol_eta_cs;
      else if (itt == 3) tol_eta = 1e-2 * tol_eta_cs;
      else tol_eta = tol_eta_cs;

      if (uh_err > 0.0) { du_max = du; max_lazy = false; }
      else if (uh_err < 0.0) { du_min = du; min_lazy = false; }
      else do_I = false;

      if (do_I) {
        if ((T.dt * IareaMin * fabs(uh_err) > tol_eta) ||
            (better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)))) {
          const double ddu = -uh_err / duhdu_tot;
          const double du_prev = du;
          du = du + ddu;
          if (fabs(ddu) < 1.0e-15 * fabs(du)) {
            do_I = false;
          } else if (ddu > 0.0) {
            if (max_lazy && !(du < du_max)) undecided = true;
            if (du >= du_max) {
              du = 0.5 * (du_prev + du_max);
              if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
            }
          } else {
            if (min_lazy && !(du > du_min)) undecided = true;
            if (du <= du_min) {
              du = 0.5 * (du_prev + du_min);
              if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
            }
          }
        } else {
          do_I = false;
        }
      }
    }
    if (T.lead) { T.s_du[T.fl] = du; T.s_doI[T.fl] = do_I ? 1 : 0; }
    const int wg = __ockl_wgred_or_i32((do_I ? 1 : 0) | (undecided ? 2 : 0));   // barrier + bitwise OR over the work-group
    if (wg & 2) { need_exact = true; break; }
    if (!(wg & 1)) break;

    if ((itt < max_itts) || store) {
      if (T.s_doI[T.fl]) {
        const double dul = T.s_du[T.fl];
#pragma unroll
        for (int n = 0; n < MAXL; n++) {
          const int k = T.kl + KL * n;
          if (k < T.nk) {
            double uh, duhdu;
            flux_lds(u_r[n] + dul * v_r[n], T.dt, T.IdT_m, T.IdT_p, v_r[n], T.Lf, T.sL, T.sR, T.sC,
                     k * T.NC + T.cm, k * T.NC + T.cp, uh, duhdu);
            T.sA[k * NF + T.fl] = uh; T.sB[k * NF + T.fl] = duhdu;
          }
        }
      }
      __syncthreads();
      if ((itt < max_itts) && do_I) {
        double err = -uhbt, dtot = 0.0;
        col_walk2(T.sA + T.fl, T.sB + T.fl, T.nk, [&](double a, double b) { err = err + a; dtot = dtot + b; });
        uh_err = err; duhdu_tot = dtot;
        uh_err_best = dmin(uh_err_best, fabs(uh_err));
      }
    }
  }
  return du;
}

template <int DIR, int KL, int MAXL, bool LAZY>
__global__ void __launch_bounds__(NF * KL, (KL * MAXL > 96) ? 1 : (DIR ? 2 : MOM6X_X_WG))   // LDS lets 3 (x) | 2 (y) work-groups share a CU at nk = 75
k_mass_flux_lds(Dm d, const double *__restrict__ G, FluxArgs A, LdsArgs E) {
  // The LAZY kernel runs flux_adjust with cheap bounds of the CFL limits; a work-group whose Newton steps come within
  // reach of them marks its tile in E.retry and stops.  The exact kernel (second launch, same grid) only works on the
  // marked tiles, with the limits from the k-recurrence, and rewrites all of the tile's outputs.
  // Tile <-> block id: block b is observed to run on XCD b % 8, each XCD with its own L2.  Every XCD gets a band of
  // E.rows tile rows; inside the band the tiles follow each other along i for the zonal kernel (neighbours share the
  // halo cells' cache lines) and along j for the meridional one (neighbours share 5 of their 6 rows of h), so the
  // re-reads hit that XCD's L2 instead of going to the fabric once per tile.  Placement is a speed matter only.
  const int tile = blockIdx.x;
  const int band = tile & 7, slot = tile >> 3;
  const int bx = DIR ? slot / E.rows : slot % E.gx;
  const int by = band * E.rows + (DIR ? slot % E.rows : slot / E.gx);
  if (by >= E.gy) return;
  if (!LAZY && E.retry && !E.retry[tile]) return;
  TICK_INIT;
  extern __shared__ double smem[];
  constexpr int NC = DIR ? 2 * NF : NF + 1;
  const int nk = d.nk, nkp = max((nk + 1) & ~1, 3 * KL);   // the transit arrays double as 6 x [KL][NF] reduction scratch
  double *sL = smem, *sR = sL + nk * NC, *sC = sR + nk * NC, *sA = sC + nk * NC, *sB = sA + nkp * NF;
  __shared__ double s_du[NF], s_duL[NF], s_duR[NF], s_red[KL][NF];
  __shared__ int s_doI[NF];

  const int tid = threadIdx.x, fl = tid % NF, kl = tid / NF;
  // the wavefront that carries the sequential walks rotates with the tile so that the walks of the
  // work-groups sharing a CU do not all queue on the same SIMD
  const int kl_face = ((bx + by) % (KL / 4)) * 4;
  // tiles start on 128-byte lines of the pitched rows (E.i_base <= A.a0): every row segment a work-group
  // reads or writes is then exactly one cache line instead of two half lines shared with its neighbours
  const int i0 = E.i_base + bx * NF, j = A.b0 + by;
  const int i = i0 + fl;
  const bool active = (i >= A.a0 && i <= A.a1);
  const bool lead = (kl == kl_face);
  const bool face = active && lead;
  const int st = DIR ? d.pitch : 1;
  const size_t slab = (size_t)d.slab;
  const DirMetrics D = dir_metrics<DIR>(G, d);
  const size_t f2 = ix2(d, active ? i : A.a1, j);   // inactive lanes alias the last face (never written)
  const double dt = A.dt;
  const bool use_visc_rem = (A.visc_rem != nullptr);
  const bool need_adjust = (A.uhbt != nullptr) || A.set_BT_cont;

  // ---- this lane's layers of u and visc_rem stay in registers ---------------------------------------
  double u_r[MAXL], v_r[MAXL];
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    const int k = kl + KL * n;
    u_r[n] = 0.0; v_r[n] = 1.0;
    if (active && k < nk) {
      const size_t f = f2 + (size_t)k * slab;
      u_r[n] = A.u[f];
      if (use_visc_rem) v_r[n] = A.visc_rem[f];
    }
  }

  // every other global load of the kernel is issued here as well: a load after the stores of uh / u_cor / h_face
  // would have to wait for those to drain (the memory counter is in order)
  const double IareaMin = dmin(D.IareaT[f2], D.IareaT[f2 + st]);
  const double uhbt_f = (A.uhbt != nullptr && face) ? A.uhbt[f2] : 0.0;
  const double dC_f = A.set_BT_cont ? D.dC[f2] : 0.0;

  // ---- PPM_reconstruction + limiter for the cells of this tile, all layers, into LDS ---------------
  if (DIR == 0) {
    // the 21 h values a layer of the tile needs (cells i0-2 .. i0+NF+2) are staged through the sA|sB space
    constexpr int NR = NF + 5;
    const double *m = D.mask2dT;
    const int g1 = i0 - 2 + fl, g2 = i0 + NF - 2 + fl;
    const bool ok1 = (g1 >= A.a0 - 2 && g1 <= A.a1 + 3), ok2 = (fl < 5) && (g2 >= A.a0 - 2 && g2 <= A.a1 + 3);
    const size_t c1 = ix2(d, ok1 ? g1 : A.a1, j), c2 = ix2(d, ok2 ? g2 : A.a1, j);
    double r1[MAXL], r2[MAXL];
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      r1[n] = 0.0; r2[n] = 0.0;
      if (k < nk) {
        const double *h = A.h_in + (size_t)k * slab;
        if (ok1) r1[n] = h[c1];
        if (ok2) r2[n] = h[c2];
      }
    }
    double mm[5], mx[5];   // masks of this lane's cell (i0+fl) and of the extra cell (i0+NF)
    const bool cell_ok = (i0 + fl >= A.a0 && i0 + fl <= A.a1 + 1);
    const bool extra_ok = (fl < MAXL) && (i0 + NF >= A.a0 && i0 + NF <= A.a1 + 1);
#pragma unroll
    for (int q = 0; q < 5; q++) {
      mm[q] = cell_ok ? m[ix2(d, i0 + fl - 2 + q, j)] : 0.0;
      mx[q] = extra_ok ? m[ix2(d, i0 + NF - 2 + q, j)] : 0.0;
    }
    double *raw = sA;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        raw[k * NR + fl] = r1[n];
        if (fl < 5) raw[k * NR + NF + fl] = r2[n];
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        double hh[5], hl = 0.0, hr = 0.0, c3 = 0.0;
#pragma unroll
        for (int q = 0; q < 5; q++) hh[q] = raw[k * NR + fl + q];
        if (cell_ok) edge5(hh, mm, E.scheme, E.monotonic, E.h_min, hl, hr, c3);
        sL[k * NC + fl] = hl; sR[k * NC + fl] = hr; sC[k * NC + fl] = c3;
        if (fl == n) {   // the 17th cell of layer k is done by lane n of the layer's 16 lanes
          hl = 0.0; hr = 0.0; c3 = 0.0;
#pragma unroll
          for (int q = 0; q < 5; q++) hh[q] = raw[k * NR + NF + q];
          if (extra_ok) edge5(hh, mx, E.scheme, E.monotonic, E.h_min, hl, hr, c3);
          sL[k * NC + NF] = hl; sR[k * NC + NF] = hr; sC[k * NC + NF] = c3;
        }
      }
    }
  } else {
    // rows j-2 .. j+3 of column i: the 5-point stencils of the cells (i,j) and (i,j+1) share 4 of them
    const double *m = D.mask2dT;
    const size_t cc = ix2(d, active ? i : A.a1, j);
    double m6[6];
#pragma unroll
    for (int q = 0; q < 6; q++) m6[q] = active ? m[cc + (size_t)(q - 2) * st] : 0.0;
    double h6[MAXL][6];
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        h6[n][q] = 0.0;
        if (active && k < nk) h6[n][q] = A.h_in[(size_t)k * slab + cc + (size_t)(q - 2) * st];
      }
    }
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        double hl = 0.0, hr = 0.0, c3 = 0.0;
        if (active) edge5(&h6[n][0], &m6[0], E.scheme, E.monotonic, E.h_min, hl, hr, c3);
        sL[k * NC + fl] = hl; sR[k * NC + fl] = hr; sC[k * NC + fl] = c3;
        hl = 0.0; hr = 0.0; c3 = 0.0;
        if (active) edge5(&h6[n][1], &m6[1], E.scheme, E.monotonic, E.h_min, hl, hr, c3);
        sL[k * NC + NF + fl] = hl; sR[k * NC + NF + fl] = hr; sC[k * NC + NF + fl] = c3;
      }
    }
  }
  __syncthreads();   // cell arrays complete; the sA|sB space is free again
  TICK(0);

  Tile T;
  T.sL = sL; T.sR = sR; T.sC = sC; T.sA = sA; T.sB = sB; T.s_du = s_du; T.s_doI = s_doI;
  T.fl = fl; T.kl = kl; T.nk = nk; T.NC = NC; T.cm = fl; T.cp = DIR ? NF + fl : fl + 1; T.lead = lead;
  T.dt = dt; T.IdT_m = D.IdT[f2]; T.IdT_p = D.IdT[f2 + st];
  T.Lf = D.Lface[f2] * 1.0;   // G%dy_Cu * por_face_areaU (== 1)

  // ---- limits on du that keep the CFL number between -1 and 1 (:646-723) ----------------------------
  double visc_rem_max = 1.0, du_max_CFL = 0.0, du_min_CFL = 0.0;
  if (need_adjust) {
    if (use_visc_rem && A.use_visc_rem_max) {   // max is order-independent: reduce over the layer lanes
      double pm = 0.0;
#pragma unroll
      for (int n = 0; n < MAXL; n++)
        if (kl + KL * n < nk) pm = dmax(pm, v_r[n]);
      s_red[kl][fl] = pm;
      __syncthreads();
      visc_rem_max = 0.0;
#pragma unroll
      for (int q = 0; q < KL; q++) visc_rem_max = dmax(visc_rem_max, s_red[q][fl]);
    }
  }
  const double CFL_dt = A.CFL_limit_adjust / dt;
  const double dx_W = D.dT[f2], dx_E = D.dT[f2 + st];
  const double maskC = D.maskC[f2];
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  // The exact limits (:646-723): with visc_rem a k-recurrence with two divisions per layer.  All lanes must call.
  auto exact_bounds = [&]() {
    du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
    du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
    recurrence4<KL, MAXL>(T, face,
      [&](int n, double *o) {
        o[0] = u_r[n]; o[1] = v_r[n];
        o[2] = (dx_W * CFL_dt - u_r[n]) / v_r[n];
        o[3] = -(dx_E * CFL_dt + u_r[n]) / v_r[n];
      },
      [&](double uk, double vrem, double q_max, double q_min) {
        if (du_max_CFL * vrem > dx_W * CFL_dt - uk * maskC) du_max_CFL = q_max;
        if (du_min_CFL * vrem < -dx_E * CFL_dt - uk * maskC) du_min_CFL = q_min;
      });
    du_max_CFL = dmax(du_max_CFL, 0.0);
    du_min_CFL = dmin(du_min_CFL, 0.0);
  };
  if (need_adjust) {
    // min_k (dx_W*CFL_dt - u_k) and min_k (dx_E*CFL_dt + u_k) over the layer lanes (order-independent, exact)
    double nmin = 1.0e300, mmin = 1.0e300;
    bool vr_ok = true;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      if (kl + KL * n < nk) {
        nmin = dmin(nmin, dx_W * CFL_dt - u_r[n]);
        mmin = dmin(mmin, dx_E * CFL_dt + u_r[n]);
        if (!(v_r[n] >= 0.0 && v_r[n] <= 1.0 + 1.0e-10)) vr_ok = false;   // (visc_rem can exceed 1 by round-off)
      }
    }
    if (!vr_ok) { nmin = -1.0; mmin = -1.0; }
    __syncthreads();                       // (s_red readers of the visc_rem_max reduction are done)
    sA[kl * NF + fl] = nmin; sB[kl * NF + fl] = mmin;
    __syncthreads();
    nmin = 1.0e300; mmin = 1.0e300;
#pragma unroll
    for (int q = 0; q < KL; q++) { nmin = dmin(nmin, sA[q * NF + fl]); mmin = dmin(mmin, sB[q * NF + fl]); }
    __syncthreads();                       // sA|sB are free again
    const double D0max = 2.0 * (CFL_dt * dx_W) * I_vrm, D0min = -2.0 * (CFL_dt * dx_E) * I_vrm;
    if (!use_visc_rem) {
      // :709-716: plain min / max chains, exact in any order
      du_max_CFL = dmax(dmin(D0max, nmin), 0.0);
      du_min_CFL = dmin(dmax(D0min, -mmin), 0.0);
    } else if (!LAZY) {
      exact_bounds();
    } else {
      // With visc_rem the recurrence leaves du_max_CFL equal to D0max or to one of q_k = (dx_W*CFL_dt - u_k)/visc_rem_k,
      // and q_k >= (1 - 1e-10)*(dx_W*CFL_dt - u_k) whenever that is >= 0 and 0 <= visc_rem_k <= 1 + 1e-10: so
      // (1 - 1e-9)*min(D0max, nmin) is a lower bound of du_max_CFL, and likewise for du_min_CFL from above.  flux_adjust is run with
      // these bounds first (see wg_flux_adjust); the recurrence itself is only needed if a Newton step reaches them.
      const double shrink = 1.0 - 1.0e-9;
      du_max_CFL = (nmin >= 0.0) ? shrink * dmax(dmin(D0max, nmin), 0.0) : -1.0e300;
      du_min_CFL = (mmin >= 0.0) ? shrink * dmin(dmax(D0min, -mmin), 0.0) : 1.0e300;
    }
  }
  const bool lazy = LAZY && use_visc_rem && (E.retry != nullptr);   // flux_adjust runs with bounds of the limits
  TICK(1);
  // The first sweep (:615-668): layer transports and their derivatives of this lane's layers into the transit arrays
  auto first_sweep = [&]() {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        double uh, duhdu;
        flux_lds(u_r[n], dt, T.IdT_m, T.IdT_p, v_r[n], T.Lf, sL, sR, sC, k * NC + T.cm, k * NC + T.cp, uh, duhdu);
        sA[k * NF + fl] = uh; sB[k * NF + fl] = duhdu;
      }
    }
  };

  // ---- first sweep: layer transports and their column sums (:615-668) -------------------------------
  first_sweep();
  double uh_tot_0 = 0.0, duhdu_tot_0 = 0.0;
  if (need_adjust) {
    __syncthreads();
    if (face)
      col_walk2(sA + fl, sB + fl, nk, [&](double a, double b) { duhdu_tot_0 = duhdu_tot_0 + b; uh_tot_0 = uh_tot_0 + a; });
  }

  TICK(2);
  // ---- flux_adjust towards uhbt; uh, u_cor, du_cor ---------------------------------------------------
  double du_fin = 0.0;
  const bool corrected = (A.uhbt != nullptr);
  if (corrected) {
    const double uhbt = uhbt_f;
    bool redo;
    const double du = wg_flux_adjust<KL, MAXL>(T, face, u_r, v_r, IareaMin, uhbt, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                               A.tol_eta, A.tol_vel, A.better_iter, true, lazy, redo);
    if (lazy && redo) { if (tid == 0) E.retry[tile] = 1; return; }   // work-group-uniform
    if (face && A.du_cor) A.du_cor[f2] = du;
    du_fin = s_du[fl];   // published by the face lane before the last barrier of the loop
  }
  TICK(3);
  if (active) {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        const size_t f = f2 + (size_t)k * slab;
        A.uh[f] = sA[k * NF + fl];
        if (corrected && A.u_cor) A.u_cor[f] = u_r[n] + du_fin * v_r[n];
      }
    }
  }

  // ---- zonal/merid_flux_thickness (:975 / :1866) at the corrected velocities ------------------------
  if (E.h_face && active) {
    const bool use_cor = corrected && (A.u_cor != nullptr);
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        const double uf = use_cor ? (u_r[n] + du_fin * v_r[n]) : u_r[n];
        const int m = k * NC + T.cm, p = k * NC + T.cp;
        double h_avg, h_marg;
        if (uf > 0.0) {
          const double CFL = uf * dt * T.IdT_m;
          const double l = sL[m], r = sR[m], curv_3 = sC[m];
          h_avg = r + CFL * (0.5 * (l - r) + curv_3 * (CFL - 1.5));
          h_marg = r + CFL * ((l - r) + 3.0 * curv_3 * (CFL - 1.0));
        } else if (uf < 0.0) {
          const double CFL = -uf * dt * T.IdT_p;
          const double l = sL[p], r = sR[p], curv_3 = sC[p];
          h_avg = l + CFL * (0.5 * (r - l) + curv_3 * (CFL - 1.5));
          h_marg = l + CFL * ((r - l) + 3.0 * curv_3 * (CFL - 1.0));
        } else {
          h_avg = 0.5 * (sL[p] + sR[m]);
          h_marg = 0.5 * (sL[p] + sR[m]);
        }
        double hu = E.marginal ? h_marg : h_avg;
        if (use_visc_rem) hu = hu * (v_r[n] * 1.0);
        else hu = hu * 1.0;
        E.h_face[f2 + (size_t)k * slab] = hu;
      }
    }
  }
  TICK(4);
  if (!A.set_BT_cont) return;

  // ---- set_zonal_BT_cont :1246-1409 / set_merid_BT_cont :2143-2304 -----------------------------------
  __syncthreads();
  const double Idt = 1.0 / dt, min_visc_rem = 0.1, CFL_min = 1e-6;
  bool redo0;
  const double du0f = wg_flux_adjust<KL, MAXL>(T, face, u_r, v_r, IareaMin, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                               A.tol_eta, A.tol_vel, A.better_iter, false, lazy, redo0);
  if (lazy && redo0) { if (tid == 0) E.retry[tile] = 1; return; }
  const double du0 = face ? du0f : 0.0;
  TICK(5);
  const double du_CFL = (CFL_min * Idt) * dC_f;
  // duR / duL (:1293-1316): a k-recurrence "if (u_k + duR*vrl_k > -du_CFL*vrem_k) duR = q_k" whose state only ever
  // holds duR_0 or one of the quotients q_k.  In exact arithmetic it returns min(duR_0, min_k q_k); in floating
  // point the comparison is made in the un-divided form and may disagree with that for near-ties.  So all lanes
  // compute the candidate F = min (first index k*) and a certificate that the reference's loop returns exactly F:
  //   - the test of layer k* is true for the smallest value s2 > F the state can hold before it (the test is monotone
  //     in the state), and
  //   - the test is false at F for every k > k*, unless q_k is bit-identical to F.
  // Only if a certificate fails (it does not unless quotients nearly tie) does the work-group walk the recurrence.
  const double a_du0 = s_du[fl];   // published by the face lane inside wg_flux_adjust
  const double x0R = dmin(0.0, a_du0 - du_CFL), x0L = dmax(0.0, a_du0 + du_CFL);
  const double vrl_floor = min_visc_rem * visc_rem_max;
  double a_duR, a_duL;
  {
    const double BIG = 1.0e300;
    double qR[MAXL], qL[MAXL], vl[MAXL];
    double r1 = BIG, r2 = BIG, l1 = -BIG, l2 = -BIG, ri = 1.0e9, li = 1.0e9;   // two smallest / largest distinct quotients
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      vl[n] = 0.0; qR[n] = 0.0; qL[n] = 0.0;
      if (k < nk) {
        const double uk = u_r[n], vrem = v_r[n];
        const double visc_rem_lim = dmax(vrem, vrl_floor);
        if (visc_rem_lim > 0.0) {
          vl[n] = visc_rem_lim;
          qR[n] = -(uk + du_CFL * vrem) / visc_rem_lim;
          qL[n] = -(uk - du_CFL * vrem) / visc_rem_lim;
          if (qR[n] < r1) { r2 = r1; r1 = qR[n]; ri = (double)k; } else if (qR[n] > r1 && qR[n] < r2) r2 = qR[n];
          if (qL[n] > l1) { l2 = l1; l1 = qL[n]; li = (double)k; } else if (qL[n] < l1 && qL[n] > l2) l2 = qL[n];
        }
      }
    }
    double *pr = sA;   // 6 x [KL][NF] <= 2 * nkp * NF
    const int o = kl * NF + fl, ps = KL * NF;
    pr[o] = r1; pr[ps + o] = ri; pr[2 * ps + o] = r2; pr[3 * ps + o] = l1; pr[4 * ps + o] = li; pr[5 * ps + o] = l2;
    __syncthreads();
    double FR = BIG, kR = 1.0e9, FL = -BIG, kL = 1.0e9;
#pragma unroll
    for (int q = 0; q < KL; q++) {
      const double v = pr[q * NF + fl], iv = pr[ps + q * NF + fl], w = pr[3 * ps + q * NF + fl], iw = pr[4 * ps + q * NF + fl];
      if (v < FR || (v == FR && iv < kR)) { FR = v; kR = iv; }
      if (w > FL || (w == FL && iw < kL)) { FL = w; kL = iw; }
    }
    double sR2 = BIG, sL2 = -BIG;
#pragma unroll
    for (int q = 0; q < KL; q++) {
      const double v = pr[q * NF + fl], v2 = pr[2 * ps + q * NF + fl], w = pr[3 * ps + q * NF + fl], w2 = pr[5 * ps + q * NF + fl];
      sR2 = dmin(sR2, (v > FR) ? v : v2);
      sL2 = dmax(sL2, (w < FL) ? w : w2);
    }
    if (FR < x0R) sR2 = dmin(sR2, x0R); else { FR = x0R; kR = -1.0; }
    if (FL > x0L) sL2 = dmax(sL2, x0L); else { FL = x0L; kL = -1.0; }
    bool bad = false;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk && vl[n] > 0.0) {
        const double uk = u_r[n], vrem = v_r[n], kd = (double)k;
        if (kd > kR) bad = bad || ((uk + FR * vl[n] > -du_CFL * vrem) && (__double_as_longlong(qR[n]) != __double_as_longlong(FR)));
        else if (kd == kR) bad = bad || !(uk + sR2 * vl[n] > -du_CFL * vrem);
        if (kd > kL) bad = bad || ((uk + FL * vl[n] < du_CFL * vrem) && (__double_as_longlong(qL[n]) != __double_as_longlong(FL)));
        else if (kd == kL) bad = bad || !(uk + sL2 * vl[n] < du_CFL * vrem);
      }
    }
    a_duR = FR; a_duL = FL;
    if (__ockl_wgred_or_i32(((bad && active) || E.force_walk) ? 1 : 0)) {   // barrier; work-group-uniform
#ifdef MOM6X_MFL_TIMING
      if (threadIdx.x == 0) atomicAdd(&g_mfl_t[DIR][15], 1ull);
#endif
      double duR = x0R, duL = x0L;
      recurrence4<KL, MAXL>(T, face,
        [&](int n, double *o4) {
          const double uk = u_r[n], vrem = v_r[n];
          const double visc_rem_lim = dmax(vrem, vrl_floor);
          o4[0] = uk; o4[1] = vrem;
          o4[2] = -(uk + du_CFL * vrem) / visc_rem_lim;
          o4[3] = -(uk - du_CFL * vrem) / visc_rem_lim;
        },
        [&](double uk, double vrem, double q_R, double q_L) {
          const double visc_rem_lim = dmax(vrem, vrl_floor);
          if (visc_rem_lim > 0.0) {
            if (uk + duR * visc_rem_lim > -du_CFL * vrem) duR = q_R;
            if (uk + duL * visc_rem_lim < du_CFL * vrem) duL = q_L;
          }
        });
      if (lead) { s_duL[fl] = duL; s_duR[fl] = duR; }
      __syncthreads();
      a_duL = s_duL[fl]; a_duR = s_duR[fl];
    }
  }
  const double duR = a_duR, duL = a_duL;
  TICK(6);
  double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
  // three trial velocities (:1330-1349), five column sums.  Every layer is evaluated once per trial velocity; the
  // five results wait in registers until all lanes are done with the cell arrays, whose space then serves as the
  // third to fifth transit array, and the face lanes run the five sums as independent chains of ONE walk.
  double t_d0[MAXL], t_dL[MAXL], t_uL[MAXL], t_uR[MAXL], t_dR[MAXL];
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    const int k = kl + KL * n;
    t_d0[n] = 0.0; t_dL[n] = 0.0; t_uL[n] = 0.0; t_uR[n] = 0.0; t_dR[n] = 0.0;
    if (k < nk) {
      double uh;
      flux_lds(u_r[n] + a_du0 * v_r[n], dt, T.IdT_m, T.IdT_p, v_r[n], T.Lf, sL, sR, sC, k * NC + T.cm, k * NC + T.cp, uh, t_d0[n]);
      flux_lds(u_r[n] + a_duL * v_r[n], dt, T.IdT_m, T.IdT_p, v_r[n], T.Lf, sL, sR, sC, k * NC + T.cm, k * NC + T.cp, t_uL[n], t_dL[n]);
      flux_lds(u_r[n] + a_duR * v_r[n], dt, T.IdT_m, T.IdT_p, v_r[n], T.Lf, sL, sR, sC, k * NC + T.cm, k * NC + T.cp, t_uR[n], t_dR[n]);
    }
  }
  __syncthreads();   // nobody reads the cell arrays any more
  double *sX = smem;   // 3 x [nk][NF] <= 3 x [nk][NC]
  const int strX = nk * NF;
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    const int k = kl + KL * n;
    if (k < nk) {
      const int o = k * NF + fl;
      sA[o] = t_d0[n]; sB[o] = t_dL[n]; sX[o] = t_uL[n]; sX[o + strX] = t_uR[n]; sX[o + 2 * strX] = t_dR[n];
    }
  }
  __syncthreads();
  TICK(7);
  if (!face) return;
  col_walk5(sA + fl, sB + fl, sX + fl, strX, nk, [&](double d0, double dL, double uL, double uR, double dR) {
    FAmt_0 = FAmt_0 + d0; FAmt_L = FAmt_L + dL; uhtot_L = uhtot_L + uL; uhtot_R = uhtot_R + uR; FAmt_R = FAmt_R + dR;
  });

  double FA_0 = FAmt_0, FA_avg = FAmt_0;
  if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
  if (FA_avg > dmax(FA_0, FAmt_L)) FA_avg = dmax(FA_0, FAmt_L);
  else if (FA_avg < dmin(FA_0, FAmt_L)) FA_0 = FA_avg;
  A.FA_m0[f2] = FA_0; A.FA_mm[f2] = FAmt_L;
  if (fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0) A.uBT_mm[f2] = 0.0;
  else A.uBT_mm[f2] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

  FA_0 = FAmt_0; FA_avg = FAmt_0;
  if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
  if (FA_avg > dmax(FA_0, FAmt_R)) FA_avg = dmax(FA_0, FAmt_R);
  else if (FA_avg < dmin(FA_0, FAmt_R)) FA_0 = FA_avg;
  A.FA_p0[f2] = FA_0; A.FA_pp[f2] = FAmt_R;
  if (fabs(FAmt_R - FA_0) <= 1e-12 * FA_0) A.uBT_pp[f2] = 0.0;
  else A.uBT_pp[f2] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));
}

template <int DIR, int KL, int MAXL>
int launch(mom6x_ctx *c, const FluxArgs &A, const LdsArgs &E0, size_t lds_bytes) {
  const Dm d = c->d;
  LdsArgs E = E0;
  E.gx = (A.a1 - E.i_base + NF) / NF; E.gy = A.b1 - A.b0 + 1;
  E.rows = (E.gy + 7) / 8;
  const dim3 grid(8 * E.gx * E.rows, 1, 1);
  const bool need_adjust = (A.uhbt != nullptr) || A.set_BT_cont;
  const bool two_pass = need_adjust && (A.visc_rem != nullptr);   // only then are the cheap bounds not the limits themselves
  E.retry = nullptr;
  if (two_pass) {
    const size_t ntile = (size_t)grid.x;
    if (c->retry_cap < ntile) {
      HIPCHK(hipStreamSynchronize(c->stream));
      (void)hipFree(c->retry);
      HIPCHK(hipMalloc(&c->retry, ntile * sizeof(int)));
      c->retry_cap = ntile;
    }
    HIPCHK(hipMemsetAsync(c->retry, 0, ntile * sizeof(int), c->stream));
    E.retry = c->retry;
  }
  auto k_lazy = k_mass_flux_lds<DIR, KL, MAXL, true>;
  auto k_exact = k_mass_flux_lds<DIR, KL, MAXL, false>;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lazy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
#ifdef MOM6X_MFL_TIMING
  {
    static bool once[2] = {false, false};
    if (!once[DIR]) {
      once[DIR] = true;
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lazy, NF * KL, lds_bytes);
      fprintf(stderr, "[mfl] dir %d: %zu B dynamic LDS, occupancy %d work-groups per CU\n", DIR, lds_bytes, nb);
    }
  }
#endif
  if (c->prof_on) prof_begin(c, DIR ? "k_mass_flux_lds<1>" : "k_mass_flux_lds<0>");
  hipLaunchKernelGGL(k_lazy, grid, dim3(NF * KL, 1, 1), lds_bytes, c->stream, d, c->G, A, E);
  if (c->prof_on) prof_end(c);
  if (two_pass) {
    if (c->prof_on) prof_begin(c, DIR ? "k_mass_flux_lds_exact<1>" : "k_mass_flux_lds_exact<0>");
    hipLaunchKernelGGL(k_exact, grid, dim3(NF * KL, 1, 1), lds_bytes, c->stream, d, c->G, A, E);
    if (c->prof_on) prof_end(c);
  }
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

}  // namespace

#ifdef MOM6X_MFL_TIMING
extern "C" int mom6x_debug_mfl_timing(unsigned long long *out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mfl_t), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfl_t), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

size_t mass_flux_lds_bytes(int dir, int nk) {
  const int NC = dir ? 2 * NF : NF + 1;
  const int nkp = std::max((nk + 1) & ~1, 48);   // the kernel's nkp (KL = 16)
  return sizeof(double) * ((size_t)nk * (size_t)(3 * NC) + (size_t)nkp * (size_t)(2 * NF));
}

bool mass_flux_lds_usable(int nk) {
  return nk <= 128 && mass_flux_lds_bytes(1, nk) <= 156 * 1024;
}

int mass_flux_lds(mom6x_ctx *c, int dir, const FluxArgs &A, const LdsArgs &E0) {
  const int nk = c->d.nk;
  LdsArgs E = E0;
  const char *env = getenv("MOM6X_MASSFLUX");
  E.force_walk = (env && !strcmp(env, "lds_walk")) ? 1 : 0;
  E.i_base = A.a0 - (((A.a0 + c->d.ioff) % NF) + NF) % NF;   // (i_base + ioff) is a multiple of 16 doubles = 128 B
  const size_t bytes = mass_flux_lds_bytes(dir, nk);
  const int maxl = (nk + 15) / 16;   // 16 layer lanes per face (32 was measured slower at nk = 75)
#define GO(D, K, M) return launch<D, K, M>(c, A, E, bytes)
  if (dir == 0) { if (maxl <= 2) GO(0, 16, 2); if (maxl <= 5) GO(0, 16, 5); GO(0, 16, 8); }
  if (maxl <= 2) GO(1, 16, 2); if (maxl <= 5) GO(1, 16, 5); GO(1, 16, 8);
#undef GO
}
