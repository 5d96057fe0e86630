// tracer.hip -- tracer advection and the vertical tridiagonal solves on gfx950.
//
//   advect_tracer / advect_x / advect_y   <- src/tracer/MOM_tracer_advect.F90:53-1153 (PLM, PPM:H3, PPM)
//   triDiagTS, triDiagTS_Eulerian         <- src/parameterizations/vertical/MOM_diabatic_aux.F90:394-488
//   tracer_vertdiff(_Eulerian)            <- src/tracer/MOM_tracer_diabatic.F90:25-420 (no-sinking branch)
//
// advect_tracer is an iteration: each pass moves as much of the remaining transport (uhr, vhr) as keeps the
// upwind cell non-empty, until nothing remains.  The reference's row flags domore_u(j,k)/domore_v(J,k) and
// layer flags domore_k(k) decide which rows are touched at all, so they are kept (int arrays in HBM) and the
// exit test reads domore_k back once per halo cycle -- the same global synchronisation point as the
// reference's sum_across_PEs (:331).  All fluxes of a pass must be formed from the un-updated hprev/uhr/tracer
// values.  Default: one kernel per direction and pass that reads everything it needs before it writes (k_ta_x_tile,
// k_ta_y_tile, see there).  MOM6X_TRACER=legacy: two kernels per direction and pass,
//   k_ta_face<DIR>: limited transport uhh and the tracer fluxes of every face   -> scratch
//   k_ta_cell<DIR>: uhr -= uhh ; hprev and every tracer updated from the face fluxes
#include <cfloat>
#include <vector>
#include "mom6x_dev.h"

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);
int comm_allreduce_int_sum(mom6x_ctx *c, int *dev, int n);   // halo.hip

#define MAXTR 8
enum { ADVECT_PLM = 0, ADVECT_PPMH3 = 1, ADVECT_PPM = 2 };

struct TAState {
  double dt_dyn; int default_scheme, useHuynhStencilBug;
  double *hprev, *uhr, *vhr, *uhh, *flux[MAXTR];
  int nflux;
  int *dmu, *dmv, *limu, *limv, *dmk;   // [nk*nrows] row flags, "a flux was limited" marks, [nk] layer flags
  double *save; size_t save_cap;        // the old values beyond the tile / segment boundaries of the one-kernel passes
  int stencil_all;                      // advect_tracer of more than MAXTR tracers: the stencil of the WHOLE registry (0: none in force)
};

struct TrList { double *t[MAXTR]; int scheme[MAXTR]; int n; };
struct FluxList { double *f[MAXTR]; };

namespace {

inline dim3 blk2() { return dim3(64, 4, 1); }

__device__ __forceinline__ double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
__device__ __forceinline__ double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }

// x / 3 and x / 6 (the PPM edge values and limiter, :560-575 / :965-980) in three operations instead of the ten of a division: with
// y = RN(1 / c), q = RN(x y) lies within one ulp of x / c, r = x - c q is exact in a fused multiply-add, and RN(q + r y) is the
// correctly rounded quotient (Markstein's theorem; y is within half an ulp of 1 / c).  The sign is the numerator's (+-0 included).
// Below 2**-1000 the quotient can be subnormal, where q + r y may meet a tie: a wavefront that holds such a value (never, in an
// ocean) divides.  Checked against the division on 3.4e9 values of every exponent, patterns around 1, 2, 4/3 and 8/3 included
// (every mismatch had a subnormal quotient); tests/test_div_by_cpu.py repeats the check on 3e7 values with the same operations on the host.
template <int C>
__device__ __forceinline__ double div_by(double x) {
  static_assert(C == 3 || C == 6, "div_by: 3 or 6");
  const double y = 1.0 / (double)C;
  // (+-Inf too: fma(-C, Inf, Inf) is a NaN where x / C is Inf; a NaN passes through either way)
  const bool odd = ((fabs(x) < 0x1p-1000) && (x != 0.0)) || (fabs(x) == __builtin_inf());
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) return x / (double)C;
  const double q = x * y;
  const double r = __builtin_fma(-(double)C, q, x);
  return copysign(__builtin_fma(r, y, q), x);
}

// PLM slope of a cell from its three values and the product of the masks of its two faces :445-449 / :824-828
// (maxima and minima through v_max_f64 / v_min_f64: they differ from (a > b) ? a : b in the sign of a zero result only -- and in which
//  operand a NaN drops out -- and every use below ends in fabs().  A NaN in tm or tp still reaches the slope through tp - tm; a NaN
//  in tc alone drops out of dMx / dMn here as it does not in the reference's max(): NaN tracers are not guaranteed to propagate
//  through the PLM slopes, but the cell's own update Tr = (Tr hlst - flux) Ihnew carries it on.)
__device__ __forceinline__ double plm_slope3(double tm, double tc, double tp, double mprod) {
  const double dMx = __builtin_fmax(__builtin_fmax(tp, tc), tm) - tc, dMn = tc - __builtin_fmin(__builtin_fmin(tp, tc), tm);
  return mprod * dsign(__builtin_fmin(__builtin_fmin(0.5 * fabs(tp - tm), 2.0 * dMx), 2.0 * dMn), tp - tm);
}

// Tracer flux through one face :546-607 / :951-1010 from gathered operands: T5 = the tracer in the upwind cell `up` and
// its two neighbours on either side (up-2 .. up+2 along the direction); mk = the face masks of the faces up-2 .. up+1
// (face n lies between the cells n and n+1).  Every variant of the advection kernels funnels into this one function, so
// they cannot differ in arithmetic.
// The limited slope of CELL c along the direction (:445-449 / :824-828): plm_slope3 of its three values with the product of the
// masks of its two faces, mC(c) * mC(c-1) (face n lies between the cells n and n+1) -- a property of the cell: every face whose
// stencil holds the cell forms the same bits, so a kernel may form it once per cell (k_ta_x_tile's slope stage).
__device__ __forceinline__ double cell_slope(double tm, double tc, double tp, double m_c, double m_cm1) { return plm_slope3(tm, tc, tp, m_c * m_cm1); }

#define DIV3(x) div_by<3>(x)
#define DIV6(x) div_by<6>(x)
// face_flux5 with the slopes of the cells up-1, up, up+1 given (sl[0..2]; PLM: sl[1] only; PPM:H3: none)
__device__ __forceinline__ double face_flux5s(int scheme, const double (&T5)[5], const double (&mk)[4], const double (&sl)[3], double uhh, double CFL) {
  const double Tm = T5[1], Tc = T5[2], Tp = T5[3];
  if (scheme == ADVECT_PPM || scheme == ADVECT_PPMH3) {
    double aL, aR;
    if (scheme == ADVECT_PPMH3) {
      aL = DIV6(5. * Tc + (2. * Tm - Tp));
      aL = dmax(dmin(Tc, Tm), aL); aL = dmin(dmax(Tc, Tm), aL);
      aR = DIV6(5. * Tc + (2. * Tp - Tm));
      aR = dmax(dmin(Tc, Tp), aR); aR = dmin(dmax(Tc, Tp), aR);
    } else {
      const double s0 = sl[0], s1 = sl[1], s2 = sl[2];
      aL = 0.5 * ((Tm + Tc) + DIV3(s0 - s1));
      aR = 0.5 * ((Tc + Tp) + DIV3(s1 - s2));
    }
    const double dA = aR - aL, mA = 0.5 * (aR + aL);
    if (mk[2] * mk[1] * (Tp - Tc) * (Tc - Tm) <= 0.) { aL = Tc; aR = Tc; }
    else {
      const double dA2_6 = DIV6(dA * dA);
      if (dA * (Tc - mA) > dA2_6) aL = (3. * Tc) - 2. * aR;
      else if (dA * (Tc - mA) < -dA2_6) aR = (3. * Tc) - 2. * aL;
    }
    const double a6 = 6. * Tc - 3. * (aR + aL);
    if (uhh >= 0.0) return uhh * (aR - 0.5 * CFL * ((aR - aL) - a6 * (1. - 2. / 3. * CFL)));
    return uhh * (aL + 0.5 * CFL * ((aR - aL) + a6 * (1. - 2. / 3. * CFL)));
  }
  const double slope = sl[1];
  if (uhh >= 0.0) return uhh * (Tc + 0.5 * slope * (1. - CFL));
  return uhh * (Tc - 0.5 * slope * (1. - CFL));
}
__device__ __forceinline__ double face_flux5(int scheme, const double (&T5)[5], const double (&mk)[4], double uhh, double CFL) {
  double sl[3] = {0., 0., 0.};
  if (scheme == ADVECT_PPM) {
    sl[0] = cell_slope(T5[0], T5[1], T5[2], mk[1], mk[0]); sl[1] = cell_slope(T5[1], T5[2], T5[3], mk[2], mk[1]);
    sl[2] = cell_slope(T5[2], T5[3], T5[4], mk[3], mk[2]);
  } else if (scheme == ADVECT_PLM) sl[1] = cell_slope(T5[1], T5[2], T5[3], mk[2], mk[1]);
  return face_flux5s(scheme, T5, mk, sl, uhh, CFL);
}

// The operands of face f gathered through pointers: T with stride stT (index f), the 2-D face mask with stride stM (f2).
__device__ __forceinline__ double face_flux(int scheme, const double *T, const double *__restrict__ mC, size_t f, size_t f2,
                                            int stT, int stM, double uhh, double CFL) {
  const size_t up = (uhh >= 0.0) ? f : f + stT, up2 = (uhh >= 0.0) ? f2 : f2 + stM;
  double T5[5], mk[4];
  T5[1] = T[up - stT]; T5[2] = T[up]; T5[3] = T[up + stT];
  mk[1] = mC[up2 - stM]; mk[2] = mC[up2];
  if (scheme == ADVECT_PPM) {
    T5[0] = T[up - 2 * stT]; T5[4] = T[up + 2 * stT];
    mk[0] = mC[up2 - 2 * stM]; mk[3] = mC[up2 + stM];
  } else { T5[0] = 0.; T5[4] = 0.; mk[0] = 0.; mk[3] = 0.; }
  return face_flux5(scheme, T5, mk, uhh, CFL);
}

// The limited transport of a face and its CFL number :490-545 / :915-950 from the remaining transports of the face and
// of its two neighbours, the (old) volumes and areas of its two cells.  Returns true if the transport was limited.
__device__ __forceinline__ bool limited_transport(double ur, double ur_m, double ur_p, double hprev_m, double hprev_p, double area_m,
                                                  double area_p, double min_h, double &uhh, double &CFL) {
  const double tiny_h = DBL_MIN;
  bool lim = false;
  if ((ur == 0.0) || ((ur < 0.0) && (hprev_p <= tiny_h)) || ((ur > 0.0) && (hprev_m <= tiny_h))) {
    uhh = 0.0; CFL = 0.0;
  } else if (ur < 0.0) {
    const double hup = hprev_p - area_p * min_h;
    const double hlos = dmax(0.0, ur_p);
    if ((((hup - hlos) + ur) < 0.0) && ((0.5 * hup + ur) < 0.0)) { uhh = dmin3(-0.5 * hup, -hup + hlos, 0.0); lim = true; }
    else uhh = ur;
    CFL = -uhh / hprev_p;
  } else {
    const double hup = hprev_m - area_m * min_h;
    const double hlos = dmax(0.0, -ur_m);
    if ((((hup - hlos) - ur) < 0.0) && ((0.5 * hup - ur) < 0.0)) { uhh = dmax3(0.5 * hup, hup - hlos, 0.0); lim = true; }
    else uhh = ur;
    CFL = uhh / hprev_m;
  }
  return lim;
}

// The update of one cell :668-711 / :1073-1125 from its old volume, the transports and tracer fluxes of its two faces.
// Returns false if the cell keeps its tracers; otherwise T_new = (T * hlst - (F_p - F_m)) * Ihnew.
template <int DIR>
__device__ __forceinline__ bool cell_update(double uh_here, double uh_m, double hprev_old, double area, double h_neglect, double &hp,
                                            double &hlst, double &Ihnew) {
  hlst = hprev_old; Ihnew = 0.0;
  hp = hlst - (uh_here - uh_m);
  if (DIR == 1) hp = dmax(hp, 0.0);
  bool do_i = true;
  if (hp <= 0.0) do_i = false;
  else if (hp < h_neglect * area) {
    hlst = hlst + (h_neglect * area - hp);
    Ihnew = 1.0 / (h_neglect * area);
  } else Ihnew = 1.0 / hp;
  return do_i && (DIR == 1 || Ihnew > 0.0);
}

// setup :170-206: remaining transports, reconstructed previous cell volume, flags.  The kernel covers the whole
// allocated plane and writes zeros outside the computational ranges, so the three work arrays need no memset first.
__global__ void __launch_bounds__(256)
k_ta_init(Dm d, const double *__restrict__ G, const double *__restrict__ h_end, const double *__restrict__ uhtr,
          const double *__restrict__ vhtr, double *__restrict__ hprev, double *__restrict__ uhr, double *__restrict__ vhr,
          int *__restrict__ dmu, int *__restrict__ dmv) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;          // position in the pitched row
  const int jp = blockIdx.y * blockDim.y + threadIdx.y;          // row of the plane
  const int k = blockIdx.z;
  const int nrow = d.slab / d.pitch;
  if (ip >= d.pitch || jp >= nrow) return;
  const int i = ip - d.ioff, j = jp - d.joff;
  const int st = d.pitch;
  const size_t c2 = (size_t)ip + (size_t)jp * (size_t)d.pitch, c = c2 + (size_t)k * d.slab;
  const bool in_i = (i >= -1 && i <= d.ni - 1), in_j = (j >= -1 && j <= d.nj - 1);
  const double ur = (in_i && in_j && j >= 0) ? uhtr[c] : 0.0, vr = (in_i && in_j && i >= 0) ? vhtr[c] : 0.0;
  uhr[c] = ur;
  vhr[c] = vr;
  {
    // The row flags of the first iteration (:242-253, k_ta_flags_eval) for the tile's own points, which no exchange changes: "is
    // any remaining transport of this row and layer non-zero" is known here, where the transports pass anyway; k_ta_flags_eval
    // then looks at the halo rows and at the columns beyond the own ones only (a wavefront is 64 points of one row: blk2()).
    const bool own = (i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1);
    const unsigned long long bu = __ballot(own && ur != 0.0), bv = __ballot(own && vr != 0.0);
    const int lane = (int)(threadIdx.x & 63), nrows = d.nj + 2 * d.halo + 1;
    if (bu && lane == __ffsll((long long)bu) - 1) dmu[k * nrows + jp] = 1;
    if (bv && lane == __ffsll((long long)bv) - 1) dmv[k * nrows + jp] = 1;
  }
  double hp = 0.0;
  if (in_i && in_j && i >= 0 && j >= 0) {
    const double aT = gm(G, d, MOM6X_G_areaT)[c2];
    hp = dmax(0.0, aT * h_end[c] + ((uhtr[c] - uhtr[c - 1]) + (vhtr[c] - vhtr[c - st])));
    hp = hp + dmax(0.0, 1.0e-13 * hp - aT * h_end[c]);
  }
  hprev[c] = hp;
}

// Re-evaluation of the row flags :242-253.  One wave per (row, layer).
template <int DIR>
__global__ void k_ta_flags_eval(Dm d, const double *__restrict__ uhr, int *__restrict__ dm, const int *__restrict__ dmk,
                                int r0, int r1, int i0, int i1, int own_done) {
  const int r = r0 + blockIdx.x, k = blockIdx.y;
  if (r > r1 || dmk[k] <= 0) return;
  const int nrows = d.nj + 2 * d.halo + 1;
  int *flag = &dm[k * nrows + r + d.joff];
  if (*flag) return;
  int any = 0;
  if (own_done && r >= 0 && r <= d.nj - 1) {   // (k_ta_init has looked at the own points of an own row: the columns beyond them are left)
    for (int i = i0 + threadIdx.x; i <= min(i1, -1); i += 64) if (uhr[ix3(d, i, r, k)] != 0.0) any = 1;
    for (int i = max(i0, d.ni) + threadIdx.x; i <= i1; i += 64) if (uhr[ix3(d, i, r, k)] != 0.0) any = 1;
  } else
  for (int i = i0 + threadIdx.x; i <= i1; i += 64) if (uhr[ix3(d, i, r, k)] != 0.0) any = 1;
  if (__any(any) && threadIdx.x == 0) *flag = 1;
}

// domore_k(k) from the row flags (:257-259, :291-294, :310-313): any(dmu[ju0..ju1]) or any(dmv[jv0..jv1]).
// mode 0: only layers with dmk > 0 are recomputed.
__global__ void k_ta_dmk(Dm d, const int *__restrict__ dmu, const int *__restrict__ dmv, int *__restrict__ dmk,
                         int ju0, int ju1, int jv0, int jv1) {
  const int k = blockIdx.x;
  if (dmk[k] <= 0) return;
  const int nrows = d.nj + 2 * d.halo + 1;
  int any = 0;
  for (int j = ju0 + threadIdx.x; j <= ju1; j += 64) if (dmu[k * nrows + j + d.joff]) any = 1;
  for (int j = jv0 + threadIdx.x; j <= jv1; j += 64) if (dmv[k * nrows + j + d.joff]) any = 1;
  const int r = __any(any) ? 1 : 0;
  if (threadIdx.x == 0) dmk[k] = r;
}

// After a direction's pass: processed rows take "was any flux limited" as their new flag.
__global__ void k_ta_flag_commit(Dm d, int *__restrict__ dm, int *__restrict__ lim, const int *__restrict__ dmk, int r0, int r1) {
  const int r = r0 + blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (r > r1 || dmk[k] <= 0) return;
  const int nrows = d.nj + 2 * d.halo + 1;
  const int idx = k * nrows + r + d.joff;
  if (dm[idx]) { dm[idx] = lim[idx]; }
  lim[idx] = 0;
}

// limited transports and tracer fluxes of the faces (a0..a1, b0..b1) :490-607 / :915-1010
template <int DIR>
__global__ void __launch_bounds__(256)
k_ta_face(Dm d, const double *__restrict__ G, const double *__restrict__ uhr, const double *__restrict__ hprev, TrList Tr,
          double *__restrict__ uhh_out, FluxList F, const int *__restrict__ dm, int *__restrict__ lim,
          const int *__restrict__ dmk, double min_h, int a0, int a1, int b0, int b1) {
  const int i = I_BASE(a0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = b0 + blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i < a0 || i > a1 || j > b1 || dmk[k] <= 0) return;
  const int st = DIR ? d.pitch : 1;
  const int nrows = d.nj + 2 * d.halo + 1;
  const size_t f2 = ix2(d, i, j), f = f2 + (size_t)k * d.slab;
  if (!dm[k * nrows + j + d.joff]) {       // a row that is not being worked on moves nothing (:1065-1067)
    uhh_out[f] = 0.0;
    for (int m = 0; m < Tr.n; m++) F.f[m][f] = 0.0;
    return;
  }
  const double *areaT = gm(G, d, MOM6X_G_areaT);
  const double *mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu);
  double uhh, CFL;
  if (limited_transport(uhr[f], uhr[f - st], uhr[f + st], hprev[f], hprev[f + st], areaT[f2], areaT[f2 + st], min_h, uhh, CFL))
    lim[k * nrows + j + d.joff] = 1;
  uhh_out[f] = uhh;
  for (int m = 0; m < Tr.n; m++) F.f[m][f] = face_flux(Tr.scheme[m], Tr.t[m], mC, f, f2, st, st, uhh, CFL);
}

// uhr -= uhh and the cell updates :668-711 / :1073-1125.  Threads run over the faces (a0..a1, b0..b1); the
// cell on the plus side of... no: thread (i,j) owns face (i,j) [its remaining transport] and cell (i,j).
template <int DIR>
__global__ void __launch_bounds__(256)
k_ta_cell(Dm d, const double *__restrict__ G, double *__restrict__ uhr, double *__restrict__ hprev, TrList Tr,
          const double *__restrict__ uhh, FluxList F, const int *__restrict__ dm, const int *__restrict__ dmk,
          double h_neglect, double H_subroundoff, int a0, int a1, int b0, int b1, int ci0, int cj0) {
  const int i = I_BASE(a0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = b0 + blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i < a0 || i > a1 || j > b1 || dmk[k] <= 0) return;
  const int st = DIR ? d.pitch : 1;
  const int nrows = d.nj + 2 * d.halo + 1;
  const size_t c2 = ix2(d, i, j), c = c2 + (size_t)k * d.slab;
  const double *areaT = gm(G, d, MOM6X_G_areaT);
  // In x only the rows being worked on are touched at all (:417); in y every face has its remaining
  // transport updated (:1073-1076) and a cell changes only if one of its faces moved something.
  if (DIR == 0 && !dm[k * nrows + j + d.joff]) return;
  const double uh_here = uhh[c];
  {
    double r = uhr[c] - uh_here;
    const double neglect = H_subroundoff * dmin(areaT[c2], areaT[c2 + st]);
    if (fabs(r) < neglect) r = 0.0;
    uhr[c] = r;
  }
  if (i < ci0 || j < cj0) return;            // the first face of the range has no cell of this range behind it
  const double uh_m = uhh[c - st];
  if ((uh_here != 0.0) || (uh_m != 0.0)) {
    double hp, hlst, Ihnew;
    const bool upd = cell_update<DIR>(uh_here, uh_m, hprev[c], areaT[c2], h_neglect, hp, hlst, Ihnew);
    hprev[c] = hp;
    if (upd) {
      for (int m = 0; m < Tr.n; m++) Tr.t[m][c] = (Tr.t[m][c] * hlst - (F.f[m][c] - F.f[m][c - st])) * Ihnew;
    }
  }
}

// ---- one kernel per direction and pass ---------------------------------------------------------------------------------
// The face and the cell kernel above exchange uhh and one flux array per tracer through HBM ((2 + 2 ntr) array passes
// per direction).  The kernels below keep them on chip.  All fluxes of a pass must be formed from the values BEFORE the
// pass, and the update is in place, so a work item may only read what it alone will overwrite:
//   x: a work-group owns TX consecutive cells of one (row, layer); everything it reads from outside them (3 tracer
//      cells, 1 volume, up to 2 transports on either side) comes from a copy k_ta_save_x made before the pass;
//   y: the stencil runs along j, so a work-group walks a segment of SEGY rows of its columns along j with the old values it
//      still needs in LDS (no neighbours in i at all); what it needs from the rows beyond its segment comes from the copy
//      of k_ta_save_y.
constexpr int TX = 240;          // cells per work-group in x (15 x 128 B: the tiles start on cache lines); thread t < TX <-> cell C0+t and its
                                 // east face, thread TX <-> the west face of the first cell, the other 15 threads only help loading
constexpr int SEGY = 128;        // rows of a work-group's segment in y
struct SaveIdx {                 // layout of one saved boundary B (first own cell / row of the part behind it)
  // per tracer m: cells B-3..B+2 at [6*m .. 6*m+5]; then hprev B-1, B; then the transports of faces B-2, B-1, B
  __host__ __device__ static int nval(int ntr) { return 6 * ntr + 5; }
  __host__ __device__ static int T(int m, int q) { return 6 * m + q; }          // q = 0..5 <-> cell B-3+q
  __host__ __device__ static int H(int ntr, int q) { return 6 * ntr + q; }      // q = 0,1 <-> cell B-1+q
  __host__ __device__ static int U(int ntr, int q) { return 6 * ntr + 2 + q; }  // q = 0..2 <-> face B-2+q
};

// x: boundaries nb = 0..ntile at the cells B = i0 + TX*nb (the last one at i1+1), rows b0..b1, all layers being worked on
__global__ void __launch_bounds__(64)
k_ta_save_x(Dm d, const double *__restrict__ uhr, const double *__restrict__ hprev, TrList Tr, const int *__restrict__ dmk,
            double *__restrict__ save, int i0, int i1, int b0, int b1, int ntile) {
  const int nb = blockIdx.x, j = b0 + blockIdx.y, k = blockIdx.z;
  if (dmk[k] <= 0) return;
  const int B = min(i0 + TX * nb, i1 + 1), nv = SaveIdx::nval(Tr.n);
  const int v = threadIdx.x;
  if (v >= nv) return;
  double *out = save + (((size_t)k * (size_t)(b1 - b0 + 1) + (size_t)(j - b0)) * (size_t)(ntile + 1) + (size_t)nb) * (size_t)nv;
  // (a scheme with a 2-point stencil never uses the outermost saved cells, which may lie beyond the row: clamp the address)
  const int off = (v < 6 * Tr.n) ? (v % 6) - 3 : ((v < 6 * Tr.n + 2) ? (v - 6 * Tr.n) - 1 : (v - 6 * Tr.n - 2) - 2);
  const int ci = min(max(B + off, -d.ioff), d.pitch - d.ioff - 1);
  const size_t a = ix3(d, ci, j, k);
  out[v] = (v < 6 * Tr.n) ? Tr.t[v / 6][a] : ((v < 6 * Tr.n + 2) ? hprev[a] : uhr[a]);
}

// Round 4: a work-group walks XR consecutive rows of its tile column.  The loads of row r+1 (this thread's cell, the saved values
// beyond the tile's ends, the row's masks and areas) are issued right after row r's values have been handed to LDS, so they travel
// while row r is computed and stored: one row and layer per work-group had every work-group pay its load latency with nothing to do
// (486 K work-groups of 16 us at 1440 x 1080 x 75).  And the limited slope of a cell is formed ONCE, by the thread that owns the cell,
// in a stage of its own (cell_slope): the three faces whose stencils hold it read it from LDS instead of forming it again.
#ifndef MOM6X_TA_XR
#define MOM6X_TA_XR 8
#endif
constexpr int XR = MOM6X_TA_XR;   // rows per work-group
template <int MAXT>
__global__ void __launch_bounds__(256)
k_ta_x_tile(Dm d, const double *__restrict__ G, double *__restrict__ uhr, double *__restrict__ hprev, TrList Tr,
            const int *__restrict__ dm, int *__restrict__ lim, const int *__restrict__ dmk, const double *__restrict__ save,
            double min_h, double h_neglect, double H_subroundoff, int i0, int i1, int b0, int b1, int ntile) {
  const int n = blockIdx.x, k = blockIdx.z;
  if (dmk[k] <= 0) return;
  const int nrows = d.nj + 2 * d.halo + 1;
  const int jA = b0 + (int)blockIdx.y * XR, jB = min(jA + XR - 1, b1);
  const int C0 = i0 + TX * n, Cend = min(C0 + TX - 1, i1), ncell = Cend - C0 + 1;
  const int t = threadIdx.x, ntr = Tr.n, nv = SaveIdx::nval(ntr);
  __shared__ double sT[MAXT][TX + 6];               // cells C0-3 .. C0+TX+2  (position p <-> cell C0-3+p)
  __shared__ double sS[MAXT][TX + 6];               // their limited slopes (positions 1 .. TX+4 are formed)
  __shared__ double s_m[TX + 6];                    // masks of the faces C0-3 .. C0+TX+2 (position p <-> face C0-3+p)
  __shared__ double s_h[TX + 2];                    // cells C0-1 .. C0+TX
  __shared__ double s_u[TX + 3];                    // faces C0-2 .. C0+TX
  __shared__ double s_uhh[TX + 1];                  // faces C0-1 .. C0+TX-1
  __shared__ double s_F[MAXT][TX + 1];
  const double *areaT = gm(G, d, MOM6X_G_areaT), *mC = gm(G, d, MOM6X_G_mask2dCu);
  const int q = (t < ncell) ? t + 1 : 0;            // this thread's face slot (slot q of the face arrays <-> face C0-1+q)
  const int e = t - ncell;                          // (t >= ncell) which of the values beyond the tile's ends this thread fetches

  // what a thread holds of a row between its load and its hand-over to LDS
  double rT[MAXT], r_h = 0., r_u = 0., r_m = 0., r_am = 0., r_ap = 0.;
  bool r_on = false;
  auto load_row = [&](int j) {
    r_on = (j <= jB) && (dm[k * nrows + j + d.joff] != 0);     // a row that is not being worked on is not touched at all (:417)
    if (!r_on) return;
    const size_t row = ix3(d, 0, j, k), row2 = ix2(d, 0, j);
    const double *svL = save + (((size_t)k * (size_t)(b1 - b0 + 1) + (size_t)(j - b0)) * (size_t)(ntile + 1) + (size_t)n) * (size_t)nv;
    const double *svR = svL + nv;                   // the boundary at Cend+1
    if (t < ncell) {
      const size_t a = row + C0 + t;
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr) rT[m] = Tr.t[m][a];
      r_h = hprev[a]; r_u = uhr[a];
    } else if (e < 3) {
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr) rT[m] = svL[SaveIdx::T(m, e)];                      // cells C0-3 .. C0-1
    } else if (e < 6) {
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr) rT[m] = svR[SaveIdx::T(m, e)];                      // cells Cend+1 .. Cend+3
    } else if (e == 6) r_h = svL[SaveIdx::H(ntr, 0)];                                                  // cell C0-1
    else if (e == 7) r_h = svR[SaveIdx::H(ntr, 1)];                                                    // cell Cend+1
    else if (e < 10) r_u = svL[SaveIdx::U(ntr, e - 8)];                                                // faces C0-2, C0-1
    else if (e == 10) r_u = svR[SaveIdx::U(ntr, 2)];                                                   // face Cend+1
    if (t < TX + 6) {   // the mask of face C0-3+t (clamped to the allocated row: a stencil that reaches beyond it does not use the value)
      const int fi = min(max(C0 - 3 + t, -d.ioff), d.pitch - d.ioff - 1);
      r_m = mC[row2 + fi];
    }
    if (t <= ncell) { const size_t f2 = row2 + (C0 - 1 + q); r_am = areaT[f2]; r_ap = areaT[f2 + 1]; }
  };

  load_row(jA);
  for (int j = jA; j <= jB; j++) {
    const bool on = r_on;                            // (uniform over the work-group)
    const double a_m = r_am, a_p = r_ap;
    if (on) {
      // ---- hand the row over to LDS
      if (t < ncell) {
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) sT[m][t + 3] = rT[m];
        s_h[t + 1] = r_h; s_u[t + 2] = r_u;
      } else if (e < 3) {
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) sT[m][e] = rT[m];
      } else if (e < 6) {
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) sT[m][ncell + e] = rT[m];
      } else if (e == 6) s_h[0] = r_h;
      else if (e == 7) s_h[ncell + 1] = r_h;
      else if (e < 10) s_u[e - 8] = r_u;
      else if (e == 10) s_u[ncell + 2] = r_u;
      if (t < TX + 6) s_m[t] = r_m;
    }
    __syncthreads();
    load_row(j + 1);                                 // ... and ask for the next row while this one is worked on
    if (on) {
      // ---- slopes of the cells C0-2 .. Cend+2 (positions 1 .. ncell+4), once per cell
      if (t < ncell + 4) {
        const int p = t + 1;
        const double m_c = s_m[p], m_cm1 = s_m[p - 1];    // faces of cell C0-3+p: face C0-3+p (east) and C0-4+p (west)
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) {
          const int sch = Tr.scheme[m];
          sS[m][p] = (sch == ADVECT_PPMH3) ? 0.0 : cell_slope(sT[m][p - 1], sT[m][p], sT[m][p + 1], m_c, m_cm1);
        }
      }
    }
    __syncthreads();
    // ---- faces C0-1 .. Cend: thread t < ncell <-> the east face C0+t of its cell, thread ncell <-> the face C0-1
    double uhh = 0.0;
    if (on && t <= ncell) {
      double CFL;
      if (limited_transport(s_u[q + 1], s_u[q], s_u[q + 2], s_h[q], s_h[q + 1], a_m, a_p, min_h, uhh, CFL))
        lim[k * nrows + j + d.joff] = 1;
      s_uhh[q] = uhh;
      const bool pos = (uhh >= 0.0);
      const int o = pos ? q : q + 1;                  // the upwind cell is at position o+2; masks of the faces up-2 .. up+1 at o .. o+3 (+2)
      const double mk[4] = {s_m[o], s_m[o + 1], s_m[o + 2], s_m[o + 3]};   // face up-2+e <-> position (o+2) - 2 + e ... see below
      for (int m = 0; m < ntr; m++) {
        const double *Tq = &sT[m][o];
        const double T5[5] = {Tq[0], Tq[1], Tq[2], Tq[3], Tq[4]};
        const double sl[3] = {sS[m][o + 1], sS[m][o + 2], sS[m][o + 3]};
        s_F[m][q] = face_flux5s(Tr.scheme[m], T5, mk, sl, uhh, CFL);
      }
    }
    __syncthreads();
    // ---- the remaining transport of the faces this work-group owns (C0 .. Cend; in the first tile of a row also i0-1),
    //      and the cells C0 .. Cend
    if (on && !(t > ncell || (t == ncell && n > 0))) {
      const size_t row = ix3(d, 0, j, k);
      const int c = C0 - 1 + q;
      {
        double r = s_u[q + 1] - uhh;
        const double neglect = H_subroundoff * dmin(a_m, a_p);
        if (fabs(r) < neglect) r = 0.0;
        uhr[row + c] = r;
      }
      if (t != ncell) {
        const double uh_m = s_uhh[q - 1];
        if ((uhh != 0.0) || (uh_m != 0.0)) {
          double hp, hlst, Ihnew;
          const bool upd = cell_update<0>(uhh, uh_m, s_h[q], a_m, h_neglect, hp, hlst, Ihnew);
          hprev[row + c] = hp;
          if (upd) for (int m = 0; m < ntr; m++) Tr.t[m][row + c] = (sT[m][q + 2] * hlst - (s_F[m][q] - s_F[m][q - 1])) * Ihnew;
        }
      }
    }
    __syncthreads();                                 // (the next row's hand-over overwrites what the update has just read)
  }
}

// y: boundaries nb = 0..nseg at the rows B = j0 + SEGY*nb (the last one at j1+1), columns i0..i1
__global__ void __launch_bounds__(256)
k_ta_save_y(Dm d, const double *__restrict__ vhr, const double *__restrict__ hprev, TrList Tr, const int *__restrict__ dmk,
            double *__restrict__ save, int i0, int i1, int j0, int j1, int nseg) {
  const int i = I_BASE(i0) + blockIdx.x * 256 + threadIdx.x, nb = blockIdx.y, k = blockIdx.z;
  if (i < i0 || i > i1 || dmk[k] <= 0) return;
  const int B = min(j0 + SEGY * nb, j1 + 1), nv = SaveIdx::nval(Tr.n), st = d.pitch;
  const size_t nx = (size_t)(i1 - i0 + 1);
  // [k][nb][value][i]: coalesced along i
  double *out = save + ((size_t)k * (size_t)(nseg + 1) + (size_t)nb) * (size_t)nv * nx + (size_t)(i - i0);
  // (rows beyond the allocated ones are only ever asked for by stencils that do not use them: clamp the address)
  auto at = [&](int r) -> size_t { return ix3(d, i, min(max(r, -d.halo), d.nj + d.halo), k); };
  for (int m = 0; m < Tr.n; m++)
    for (int q = 0; q < 6; q++) out[(size_t)SaveIdx::T(m, q) * nx] = Tr.t[m][at(B - 3 + q)];
  for (int q = 0; q < 2; q++) out[(size_t)SaveIdx::H(Tr.n, q) * nx] = hprev[at(B - 1 + q)];
  for (int q = 0; q < 3; q++) out[(size_t)SaveIdx::U(Tr.n, q) * nx] = vhr[at(B - 2 + q)];
  (void)st;
}

// y, round 5: the zonal kernel's shape turned by a quarter.  A work-group owns YC columns of a segment of SEGY rows and walks it in
// blocks of YR rows, thread (c, r) <-> face Jb + r (the north face of cell Jb + r) of column c.  The OLD values the stencils need live
// in LDS in rings of YG = YR + 5 rows (row q in slot q mod YG): a block brings in YR new rows of every array (T rows Jb+3 .. Jb+YR+2,
// masks of the faces Jb+2 .., volumes / transports / areas of the rows Jb+1 ..), everything below them is still there from the block
// before -- rows this work-group has overwritten in memory since.  The next block's rows are asked for (into registers: 4 + MAXT
// doubles per thread) right after this block's have been handed over, so they travel while the block is computed; the limited slope
// of a cell is formed once (cell_slope), not by each of the three faces whose stencil holds it.  The transport and the fluxes of a
// block's last face wait in row 0 of s_uhh / s_F for the first cell of the next block.  A "block" of staging only (Jb = R0 - 1 - YR)
// fills the rings before the first faces.  What lies beyond the segment comes from k_ta_save_y's copy.  (Rounds 2-4 had one thread
// march along j per column with its window in registers: 4.4 ms per pass of four tracers against 4.3 here, profiles/r05_tracer.md.)
constexpr int YR = 8, YG = YR + 5, YCW = 32;   // rows per block, rows of a ring, columns of a work-group (up to four tracers; 16 with more)
template <int MAXT, int YC>
__global__ void __launch_bounds__(256)
k_ta_y_tile(Dm d, const double *__restrict__ G, double *__restrict__ vhr, double *__restrict__ hprev, TrList Tr,
            const int *__restrict__ dm, int *__restrict__ lim, const int *__restrict__ dmk, const double *__restrict__ save,
            double min_h, double h_neglect, double H_subroundoff, int i0, int i1, int j0, int j1, int nseg) {
  const int n = blockIdx.y, k = blockIdx.z;
  if (dmk[k] <= 0) return;
  constexpr int YP = YC + 1;                         // (padded rows)
  const int c = (int)threadIdx.x % YC, r = (int)threadIdx.x / YC;
  const int i_raw = I_BASE(i0) + (int)blockIdx.x * YC + c;
  const bool col_on = (i_raw >= i0 && i_raw <= i1);
  const int i = min(max(i_raw, i0), i1);             // (a column outside the range reads its neighbour's values and stores nothing)
  const int nrows = d.nj + 2 * d.halo + 1, st = d.pitch, ntr = Tr.n, nv = SaveIdx::nval(ntr);
  const int R0 = j0 + SEGY * n, R1 = min(R0 + SEGY - 1, j1);
  const size_t nx = (size_t)(i1 - i0 + 1);
  const double *svL = save + ((size_t)k * (size_t)(nseg + 1) + (size_t)n) * (size_t)nv * nx + (size_t)(i - i0);
  const double *svR = svL + (size_t)nv * nx;
  const double *areaT = gm(G, d, MOM6X_G_areaT), *mC = gm(G, d, MOM6X_G_mask2dCv);
  const size_t col = ix3(d, i, 0, k), col2 = ix2(d, i, 0);
  auto row2 = [&](int q) -> size_t { return col2 + (size_t)((long)min(max(q, -d.halo), d.nj + d.halo) * st); };   // clamped 2-D address
  auto row3 = [&](int q) -> size_t { return col + (size_t)((long)q * st); };
  __shared__ double sT[MAXT][YG][YP];                // old tracer values, rows Jb-2 .. Jb+YR+2
  __shared__ double s_h[YG][YP], s_v[YG][YP], s_a[YG][YP];   // volumes / areas of the rows Jb .. Jb+YR, transports of the faces Jb-1 .. Jb+YR
  __shared__ double s_m[YG][YP];                     // masks of the faces Jb-2 .. Jb+YR+1
  __shared__ double s_uhh[YR + 2][YP], s_F[MAXT][YR + 2][YP];   // row q+1 <-> face Jb+q; rows 0 and YR+1 in turn <-> the last face of the block before
  auto slot = [](int x) { x = (x >= 2 * YG) ? x - 2 * YG : x; return (x >= YG) ? x - YG : x; };   // x mod YG for 0 <= x < 3 YG
  auto nxt = [](int s) { return (s + 1 == YG) ? 0 : s + 1; };

  // what a thread holds of the block after this one between its loads and their hand-over to LDS
  double rT[MAXT], r_h = 0., r_v = 0., r_a = 0., r_m = 0.;
#pragma unroll
  for (int m = 0; m < MAXT; m++) rT[m] = 0.;
  auto load_block = [&](int JB) {
    const int qT = JB + 3 + r, qH = JB + 1 + r, qM = JB + 2 + r;
    if (qT >= R0 - 3 && qT <= R1 + 3) {
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr)
        rT[m] = (qT < R0) ? svL[(size_t)SaveIdx::T(m, qT - (R0 - 3)) * nx]
                          : ((qT > R1) ? svR[(size_t)SaveIdx::T(m, 3 + (qT - R1 - 1)) * nx] : Tr.t[m][row3(qT)]);
    }
    if (qH >= R0 - 2 && qH <= R1 + 1)
      r_v = (qH < R0) ? svL[(size_t)SaveIdx::U(ntr, qH - (R0 - 2)) * nx] : ((qH > R1) ? svR[(size_t)SaveIdx::U(ntr, 2) * nx] : vhr[row3(qH)]);
    if (qH >= R0 - 1 && qH <= R1 + 1) {
      r_h = (qH < R0) ? svL[(size_t)SaveIdx::H(ntr, 0) * nx] : ((qH > R1) ? svR[(size_t)SaveIdx::H(ntr, 1) * nx] : hprev[row3(qH)]);
      r_a = areaT[row2(qH)];
    }
    if (qM >= R0 - 3 && qM <= R1 + 2) r_m = mC[row2(qM)];
  };

  int sJb = 0, cr = 0;                               // the ring slot of row Jb; the row of s_uhh / s_F that holds the block before's last face
  load_block(R0 - 1 - YR);
  for (int Jb = R0 - 1 - YR; Jb <= R1; Jb += YR, sJb = slot(sJb + YR), cr = YR + 1 - cr) {
    // ---- hand the new rows over to LDS
    {
      const int qT = Jb + 3 + r, qH = Jb + 1 + r, qM = Jb + 2 + r;
      if (qT >= R0 - 3 && qT <= R1 + 3) {
        const int s = slot(sJb + 3 + r + YG);
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) sT[m][s][c] = rT[m];
      }
      const int sH = slot(sJb + 1 + r + YG);
      if (qH >= R0 - 2 && qH <= R1 + 1) s_v[sH][c] = r_v;
      if (qH >= R0 - 1 && qH <= R1 + 1) { s_h[sH][c] = r_h; s_a[sH][c] = r_a; }
      if (qM >= R0 - 3 && qM <= R1 + 2) s_m[slot(sJb + 2 + r + YG)][c] = r_m;
    }
    __syncthreads();
    if (Jb + YR <= R1) load_block(Jb + YR);          // ... and ask for the next block's while this one is worked on
    if (Jb < R0 - 1) continue;                        // (the staging block)
    // ---- face J = Jb + r from the old values
    const int J = Jb + r;
    const bool face_on = (J <= R1);
    const int sj = slot(sJb + r + YG);               // the slot of row J
    double uhh = 0.0, Fl[MAXT], Tc[MAXT], h_m = 0., a_m = 0.;
#pragma unroll
    for (int m = 0; m < MAXT; m++) { Fl[m] = 0.0; Tc[m] = 0.0; }
    if (face_on) {
      const int sjm = slot(sJb + r - 1 + YG), sjp = nxt(sj);
      const double v_c = s_v[sj][c], a_p = s_a[sjp][c];
      h_m = s_h[sj][c]; a_m = s_a[sj][c];
      if (dm[k * nrows + J + d.joff]) {     // a row of faces that is not being worked on moves nothing (:1065-1067)
        double CFL;
        if (limited_transport(v_c, s_v[sjm][c], s_v[sjp][c], h_m, s_h[sjp][c], a_m, a_p, min_h, uhh, CFL) && col_on) lim[k * nrows + J + d.joff] = 1;
        const bool pos = (uhh >= 0.0);
        int su[5];                                   // slots of the rows up-2 .. up+2 (up = the upwind cell: J or J+1)
        su[0] = slot(sJb + r - 2 + (pos ? 0 : 1) + YG);
#pragma unroll
        for (int q = 1; q < 5; q++) su[q] = nxt(su[q - 1]);
        const double mk[4] = {s_m[su[0]][c], s_m[su[1]][c], s_m[su[2]][c], s_m[su[3]][c]};   // faces up-2 .. up+1
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) {
          const double T5[5] = {sT[m][su[0]][c], sT[m][su[1]][c], sT[m][su[2]][c], sT[m][su[3]][c], sT[m][su[4]][c]};
          Fl[m] = face_flux5(Tr.scheme[m], T5, mk, uhh, CFL);
        }
      }
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr) Tc[m] = sT[m][sj][c];
      // the face's remaining transport (every face, :1073-1076); the first face of a segment belongs to the segment before it
      if (col_on && (J >= R0 || n == 0)) {
        double rr = v_c - uhh;
        const double neglect = H_subroundoff * dmin(a_m, a_p);
        if (fabs(rr) < neglect) rr = 0.0;
        vhr[row3(J)] = rr;
      }
      s_uhh[r + 1][c] = uhh;
#pragma unroll
      for (int m = 0; m < MAXT; m++) if (m < ntr) s_F[m][r + 1][c] = Fl[m];
      if (r == YR - 1) {                              // ... and for the first cell of the next block (which reads the other row)
        s_uhh[YR + 1 - cr][c] = uhh;
#pragma unroll
        for (int m = 0; m < MAXT; m++) if (m < ntr) s_F[m][YR + 1 - cr][c] = Fl[m];
      }
    }
    __syncthreads();
    // ---- cell J (between the faces J-1 and J)
    if (face_on && col_on && J >= R0) {
      const int rm = (r == 0) ? cr : r;
      const double uh_m = s_uhh[rm][c];
      if ((uhh != 0.0) || (uh_m != 0.0)) {
        double hp, hlst, Ihnew;
        const bool upd = cell_update<1>(uhh, uh_m, h_m, a_m, h_neglect, hp, hlst, Ihnew);
        const size_t cc = row3(J);
        hprev[cc] = hp;
        if (upd) {
#pragma unroll
          for (int m = 0; m < MAXT; m++) if (m < ntr) Tr.t[m][cc] = (Tc[m] * hlst - (Fl[m] - s_F[m][rm][c])) * Ihnew;
        }
      }
    }
    // (no barrier here: the cells read registers and s_uhh / s_F only, which nobody writes before the next two barriers; every ring
    //  row the next hand-over overwrites was last read by the faces, before the barrier above)
  }
}

// triDiagTS :394-440 / triDiagTS_Eulerian :444-488 / tracer_vertdiff(_Eulerian) no-sink branch, one column per thread.
// vertdiff = 1 adds the land mask, ea(1) in the first denominator and the surface/bottom sources.
__global__ void __launch_bounds__(256)
k_tridiag(Dm d, const double *__restrict__ G, const double *__restrict__ hold, const double *__restrict__ ea,
          const double *__restrict__ eb, double *__restrict__ T, double *__restrict__ S, double *__restrict__ c1, double h_neglect,
          int vertdiff, const double *__restrict__ sfc_flux, const double *__restrict__ btm_flux, double flux_scale, int i0, int i1,
          int j0, int j1) {
  // S != null: triDiagTS solves for T and S with the same b1, c1, d1 (:411-438) -- one sweep for both
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > i1 || j > j1) return;
  const int nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const bool two = (S != nullptr);
  double sfc_src = 0.0, btm_src = 0.0;
  if (vertdiff) {
    if (!(gm(G, d, MOM6X_G_mask2dT)[x] > 0.0)) return;
    if (sfc_flux) sfc_src = (flux_scale != 0.0) ? (sfc_flux[x] * flux_scale) : sfc_flux[x];
    if (btm_flux) btm_src = (flux_scale != 0.0) ? (btm_flux[x] * flux_scale) : btm_flux[x];
  }
  double h_tr = hold[x] + h_neglect;
  double b1 = vertdiff ? 1.0 / ((h_tr + ea[x]) + eb[x]) : 1.0 / (h_tr + eb[x]);
  double d1 = h_tr * b1;
  double prev = (b1 * h_tr) * T[x], prevS = 0.0;
  if (vertdiff) prev = prev + b1 * sfc_src;
  T[x] = prev;
  if (two) { prevS = (b1 * h_tr) * S[x]; S[x] = prevS; }
  for (int k = 1; k < nz; k++) {
    const size_t c = x + (size_t)k * slab;
    c1[c] = eb[c - slab] * b1;
    h_tr = hold[c] + h_neglect;
    const double eak = ea[c];
    const double b_denom_1 = h_tr + d1 * eak;
    b1 = 1.0 / (b_denom_1 + eb[c]);
    d1 = b_denom_1 * b1;
    if (vertdiff && k == nz - 1) prev = b1 * ((h_tr * T[c] + btm_src) + eak * prev);
    else prev = b1 * (h_tr * T[c] + eak * prev);
    T[c] = prev;
    if (two) { prevS = b1 * (h_tr * S[c] + eak * prevS); S[c] = prevS; }
  }
  for (int k = nz - 2; k >= 0; k--) {
    const size_t c = x + (size_t)k * slab;
    const double c1k = c1[c + slab];
    prev = T[c] + c1k * prev;
    T[c] = prev;
    if (two) { prevS = S[c] + c1k * prevS; S[c] = prevS; }
  }
}

// tracer_vertdiff with sink_rate :123-179 (and tracer_vertdiff_Eulerian's :315-380 with ea = ent(K), eb = ent(K+1)): one column per
// thread.  The limited sinking distances are a bottom-up recurrence and the solve runs top-down, so the first sweep leaves sink(K)
// and h_minus_dsink(k) in two scratch arrays (a tracer package's call, not on the benchmark's path: the plain form of k_tridiag).
__global__ void __launch_bounds__(256)
k_tridiag_sink(Dm d, const double *__restrict__ G, const double *__restrict__ hold, const double *__restrict__ ea,
               const double *__restrict__ eb, double *__restrict__ T, double *__restrict__ c1, double *__restrict__ snk, double *__restrict__ hmd,
               double h_neglect, const double *__restrict__ sfc_flux, const double *__restrict__ btm_flux, double flux_scale,
               double *__restrict__ btm_reservoir, double sink_dist, double H_to_RZ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const int nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  if (!(gm(G, d, MOM6X_G_mask2dT)[x] > 0.0)) return;      // (the reference forms sink on land too; nothing there reads it)
  double sfc_src = 0.0, btm_src = 0.0;
  if (sfc_flux) sfc_src = (flux_scale != 0.0) ? (sfc_flux[x] * flux_scale) : sfc_flux[x];
  if (btm_flux) btm_src = (flux_scale != 0.0) ? (btm_flux[x] * flux_scale) : btm_flux[x];
  // ---- sinking distances at the interfaces K = nz .. 1 (snk[k] <-> the top of layer k) :127-149
  double s_below = btm_reservoir ? sink_dist : 0.0;        // sink(nz+1)
  const double s_bottom = s_below;
  for (int k = nz - 1; k >= 1; k--) {
    const size_t c = x + (size_t)k * slab;
    const double h = hold[c];
    double sk, hm;
    if (btm_reservoir) { sk = sink_dist; hm = h; }
    else if (s_below >= sink_dist) { sk = sink_dist; hm = h + (s_below - sk); }
    else if (s_below + h < sink_dist) { sk = s_below + h; hm = 0.0; }
    else { sk = sink_dist; hm = (h + s_below) - sk; }
    snk[c] = sk; hmd[c] = hm;
    s_below = sk;
  }
  // ---- the solve :155-179
  double b_denom_1 = (hold[x] + s_below) + ea[x] + h_neglect;   // h_minus_dsink(1) = h_old(1) + sink(2)
  double b1 = 1.0 / (b_denom_1 + eb[x]);
  double d1 = b_denom_1 * b1;
  double h_tr = hold[x] + h_neglect;
  double prev = (b1 * h_tr) * T[x] + b1 * sfc_src;
  T[x] = prev;
  for (int k = 1; k < nz; k++) {
    const size_t c = x + (size_t)k * slab;
    c1[c] = eb[c - slab] * b1;
    const double es = ea[c] + snk[c];
    b_denom_1 = hmd[c] + d1 * es + h_neglect;
    b1 = 1.0 / (b_denom_1 + eb[c]);
    d1 = b_denom_1 * b1;
    h_tr = hold[c] + h_neglect;
    if (k == nz - 1) prev = b1 * ((h_tr * T[c] + btm_src) + es * prev);
    else prev = b1 * (h_tr * T[c] + es * prev);
    T[c] = prev;
  }
  if (btm_reservoir) btm_reservoir[x] = btm_reservoir[x] + (s_bottom * prev) * H_to_RZ;
  for (int k = nz - 2; k >= 0; k--) {
    const size_t c = x + (size_t)k * slab;
    prev = T[c] + c1[c + slab] * prev;
    T[c] = prev;
  }
}

}  // namespace

void ta_state_free(mom6x_ctx *c) {
  TAState *s = (TAState *)c->ta;
  if (!s) return;
  (void)hipFree(s->hprev); (void)hipFree(s->uhr); (void)hipFree(s->vhr); (void)hipFree(s->uhh);
  for (int m = 0; m < MAXTR; m++) (void)hipFree(s->flux[m]);
  (void)hipFree(s->dmu); (void)hipFree(s->dmv); (void)hipFree(s->limu); (void)hipFree(s->limv); (void)hipFree(s->dmk);
  (void)hipFree(s->save);
  delete s;
  c->ta = nullptr;
}

// tracer_advect_init :1155 (DT, TRACER_ADVECTION_SCHEME, USE_HUYNH_STENCIL_BUG)
extern "C" int mom6x_tracer_advect_init(mom6x_ctx *c, double dt_dyn, int default_scheme, int useHuynhStencilBug) {
  REQUIRE(c && dt_dyn > 0.0, MOM6X_EINVAL, "mom6x_tracer_advect_init: bad arguments");
  REQUIRE(default_scheme >= 0 && default_scheme <= 2, MOM6X_EINVAL, "tracer_advect_init: unknown TRACER_ADVECTION_SCHEME");
  HIPCHK(hipSetDevice(c->device));
  if (!c->ta) {
    TAState *s = new TAState();
    memset(s, 0, sizeof(*s));
    const size_t n3 = (size_t)c->dims.slab * c->dims.nk;
    const size_t nf = (size_t)(c->dims.nj + 2 * c->dims.halo + 1) * c->dims.nk;
    double **p3[] = { &s->hprev, &s->uhr, &s->vhr };   // (uhh and the flux arrays belong to the legacy path: allocated on its first use)
    for (double **q : p3) { HIPCHK(hipMalloc(q, n3 * sizeof(double))); HIPCHK(hipMemsetAsync(*q, work_fill_byte(), n3 * sizeof(double), c->stream)); }
    int **pf[] = { &s->dmu, &s->dmv, &s->limu, &s->limv };
    for (int **q : pf) { HIPCHK(hipMalloc(q, nf * sizeof(int))); HIPCHK(hipMemsetAsync(*q, 0, nf * sizeof(int), c->stream)); }
    HIPCHK(hipMalloc(&s->dmk, c->dims.nk * sizeof(int)));
    c->ta = s;
  }
  TAState *s = (TAState *)c->ta;
  s->dt_dyn = dt_dyn; s->default_scheme = default_scheme; s->useHuynhStencilBug = useHuynhStencilBug;
  return MOM6X_OK;
}

// advect_tracer :53
extern "C" int mom6x_advect_tracer(mom6x_ctx *c, const double *h_end, const double *uhtr, const double *vhtr, double dt,
                                   double *const *tracers, const int *schemes, int ntr, int x_first_in, int max_iter_in,
                                   double *uhr_out, double *vhr_out, int *iters_out) {
  REQUIRE(c && c->ta, MOM6X_EINVAL, "MOM_tracer_advect: tracer_advect_init must be called before advect_tracer.");
  REQUIRE(h_end && uhtr && vhtr && (ntr == 0 || tracers), MOM6X_EINVAL, "advect_tracer: null array");
  if (ntr == 0) return MOM6X_OK;
  if (ntr > MAXTR) {
    // The reference's tracer registry has no upper bound.  The kernels carry up to MAXTR tracers through one pass over hprev and
    // the remaining transports; a longer list is advected MAXTR at a time, each group through the whole iteration from the same
    // h_end, uhtr, vhtr: the evolution of hprev, uhr, vhr and of the row / layer flags does not depend on the tracers, so every
    // tracer gets the bits it would get in one pass (and the groups' leftover transports are identical).
    // The stencil (:118-127) is the registry's, not a group's: a PLM-only group next to a PPM one must walk the same work ranges, halo
    // cadence and iteration count as the single pass would (every group then does, and the iteration count is the same for all).
    TAState *s0 = (TAState *)c->ta;
    int st_all = 2;
    for (int m = 0; m < ntr; m++) {
      const int sch = (schemes && schemes[m] >= 0) ? schemes[m] : s0->default_scheme;
      int sl = 2;
      if (sch == ADVECT_PPM) sl = 3;
      else if (sch == ADVECT_PPMH3) sl = s0->useHuynhStencilBug ? 2 : 3;
      if (sl > st_all) st_all = sl;
    }
    int rc = MOM6X_OK, it_max = 0;
    for (int m0 = 0; m0 < ntr && rc == MOM6X_OK; m0 += MAXTR) {
      const int n = (ntr - m0 < MAXTR) ? (ntr - m0) : MAXTR;
      const bool last = (m0 + n >= ntr);
      int it = 0;
      s0->stencil_all = st_all;
      rc = mom6x_advect_tracer(c, h_end, uhtr, vhtr, dt, tracers + m0, schemes ? schemes + m0 : nullptr, n, x_first_in, max_iter_in,
                               last ? uhr_out : nullptr, last ? vhr_out : nullptr, &it);
      s0->stencil_all = 0;
      if (it > it_max) it_max = it;
    }
    if (iters_out) *iters_out = it_max;
    return rc;
  }
  HIPCHK(hipSetDevice(c->device));
  TAState *s = (TAState *)c->ta;
  const Dm d = c->d;
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1, nz = d.nk, w = d.halo;
  const size_t n3 = (size_t)d.slab * nz;
  const int nrows = d.nj + 2 * d.halo + 1;
  const size_t nf = (size_t)nrows * nz;
  hipStream_t st = c->stream;
  TrList Tr; Tr.n = ntr;
  int stencil = 2;
  for (int m = 0; m < ntr; m++) {
    Tr.t[m] = tracers[m];
    Tr.scheme[m] = (schemes && schemes[m] >= 0) ? schemes[m] : s->default_scheme;
    REQUIRE(Tr.scheme[m] >= 0 && Tr.scheme[m] <= 2, MOM6X_EINVAL, "advect_tracer: unknown advection scheme");
    int sl = 2;
    if (Tr.scheme[m] == ADVECT_PPM) sl = 3;
    else if (Tr.scheme[m] == ADVECT_PPMH3) sl = s->useHuynhStencilBug ? 2 : 3;
    if (sl > stencil) stencil = sl;
  }
  if (s->stencil_all > stencil) stencil = s->stencil_all;   // (one group of a longer registry: see above)
  REQUIRE(w >= stencil, MOM6X_EINVAL, "MOM_tracer_advect: stencil is wider than the halo.");
  // MOM6X_TRACER=legacy: the two-kernel passes that exchange uhh and the tracer fluxes through HBM
  const char *env = getenv("MOM6X_TRACER");
  const bool legacy = (env && !strcmp(env, "legacy"));
  FluxList F;
  for (int m = 0; m < MAXTR; m++) F.f[m] = nullptr;
  for (int m = 0; legacy && m < ntr; m++) {
    if (!s->flux[m]) { HIPCHK(hipMalloc(&s->flux[m], n3 * sizeof(double))); HIPCHK(hipMemsetAsync(s->flux[m], 0, n3 * sizeof(double), st)); }
    F.f[m] = s->flux[m];
  }
  bool x_first = ((c->first_direction % 2) == 0);
  int max_iter = 2 * (int)ceil(dt / s->dt_dyn) + 1;
  if (max_iter_in > 0) max_iter = max_iter_in;
  if (x_first_in >= 0) x_first = (x_first_in != 0);
  const dim3 b = blk2();
  const double min_h = 0.1 * c->GV.Angstrom_H, h_neglect = c->GV.H_subroundoff;

  if (legacy) {
    if (!s->uhh) HIPCHK(hipMalloc(&s->uhh, n3 * sizeof(double)));
    HIPCHK(hipMemsetAsync(s->uhh, 0, n3 * sizeof(double), st));
  }
  for (int *p : { s->dmu, s->dmv, s->limu, s->limv }) HIPCHK(hipMemsetAsync(p, 0, nf * sizeof(int), st));
  std::vector<int> ones(nz, 1), dmk_h(nz, 1);
  HIPCHK(hipMemcpyAsync(s->dmk, ones.data(), nz * sizeof(int), hipMemcpyHostToDevice, st));
  KLAUNCH(c, "k_ta_init", k_ta_init, grid3(d.pitch, d.slab / d.pitch, nz, b), b, d, c->G, h_end, uhtr, vhtr, s->hprev, s->uhr, s->vhr, s->dmu, s->dmv);

  int rc_tile = MOM6X_OK;
  auto need_save = [&](size_t n) -> int {
    if (s->save_cap < n) {
      if (s->save) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(s->save)); s->save = nullptr; s->save_cap = 0; }
      HIPCHK(hipMalloc(&s->save, n * sizeof(double)));
      s->save_cap = n;
    }
    return MOM6X_OK;
  };
  // one kernel per direction and pass (+ the copy of the old values beyond the tile / segment boundaries)
  auto advect_tiled = [&](int dir, int i0, int i1, int j0, int j1) -> int {
    const int nv = SaveIdx::nval(ntr);
    if (dir == 0) {
      const int ntile = (i1 - i0 + TX) / TX, nrw = j1 - j0 + 1;
      int rc = need_save((size_t)nz * nrw * (ntile + 1) * nv); if (rc) return rc;
      KLAUNCH(c, "k_ta_save_x", k_ta_save_x, dim3(ntile + 1, nrw, nz), dim3(64), d, (const double *)s->uhr, (const double *)s->hprev, Tr,
              (const int *)s->dmk, s->save, i0, i1, j0, j1, ntile);
#define XT(M) KLAUNCH(c, "k_ta_x_tile", k_ta_x_tile<M>, dim3(ntile, (nrw + XR - 1) / XR, nz), dim3(256), d, c->G, s->uhr, s->hprev, Tr, (const int *)s->dmu, \
                      s->limu, (const int *)s->dmk, (const double *)s->save, min_h, h_neglect, c->GV.H_subroundoff, i0, i1, j0, j1, ntile)
      if (ntr <= 2) XT(2); else if (ntr <= 4) XT(4); else XT(8);
#undef XT
      KLAUNCH(c, "k_ta_flag_commit", k_ta_flag_commit, dim3((j1 - j0 + 64) / 64, nz), dim3(64), d, s->dmu, s->limu, (const int *)s->dmk, j0, j1);
    } else {
      const int nseg = (j1 - j0 + SEGY) / SEGY, nx = i1 - i0 + 1, gx = (nxa(nx, i0) + 255) / 256;
      int rc = need_save((size_t)nz * (nseg + 1) * nv * nx); if (rc) return rc;
      KLAUNCH(c, "k_ta_save_y", k_ta_save_y, dim3(gx, nseg + 1, nz), dim3(256), d, (const double *)s->vhr, (const double *)s->hprev, Tr,
              (const int *)s->dmk, s->save, i0, i1, j0, j1, nseg);
#define YT(M, C) KLAUNCH(c, "k_ta_y_tile", (k_ta_y_tile<M, C>), dim3((nxa(nx, i0) + C - 1) / C, nseg, nz), dim3(C * YR), d, c->G, s->vhr, s->hprev, Tr, \
                         (const int *)s->dmv, s->limv, (const int *)s->dmk, (const double *)s->save, min_h, h_neglect, c->GV.H_subroundoff, i0, i1, j0, j1, nseg)
      if (ntr <= 2) YT(2, YCW); else if (ntr <= 4) YT(4, YCW); else YT(8, 16);
#undef YT
      KLAUNCH(c, "k_ta_flag_commit", k_ta_flag_commit, dim3((j1 - (j0 - 1) + 64) / 64, nz), dim3(64), d, s->dmv, s->limv, (const int *)s->dmk, j0 - 1, j1);
    }
    return MOM6X_OK;
  };
  auto advect = [&](int dir, int i0, int i1, int j0, int j1) {
    if (!legacy) { const int rc = advect_tiled(dir, i0, i1, j0, j1); if (rc && !rc_tile) rc_tile = rc; return; }
    // faces: x: (i0-1..i1, j0..j1); y: (i0..i1, j0-1..j1)
    const int a0 = dir ? i0 : i0 - 1, b0 = dir ? j0 - 1 : j0;
    const dim3 g = grid3(nxa(i1 - a0 + 1, a0), j1 - b0 + 1, nz, b);
    if (dir == 0) {
      KLAUNCH(c, "k_ta_face<0>", k_ta_face<0>, g, b, d, c->G, (const double *)s->uhr, (const double *)s->hprev, Tr, s->uhh, F,
              (const int *)s->dmu, s->limu, (const int *)s->dmk, min_h, a0, i1, b0, j1);
      KLAUNCH(c, "k_ta_cell<0>", k_ta_cell<0>, g, b, d, c->G, s->uhr, s->hprev, Tr, (const double *)s->uhh, F, (const int *)s->dmu,
              (const int *)s->dmk, h_neglect, c->GV.H_subroundoff, a0, i1, b0, j1, i0, j0);
      KLAUNCH(c, "k_ta_flag_commit", k_ta_flag_commit, dim3((j1 - j0 + 64) / 64, nz), dim3(64), d, s->dmu, s->limu, (const int *)s->dmk, j0, j1);
    } else {
      KLAUNCH(c, "k_ta_face<1>", k_ta_face<1>, g, b, d, c->G, (const double *)s->vhr, (const double *)s->hprev, Tr, s->uhh, F,
              (const int *)s->dmv, s->limv, (const int *)s->dmk, min_h, a0, i1, b0, j1);
      KLAUNCH(c, "k_ta_cell<1>", k_ta_cell<1>, g, b, d, c->G, s->vhr, s->hprev, Tr, (const double *)s->uhh, F, (const int *)s->dmv,
              (const int *)s->dmk, h_neglect, c->GV.H_subroundoff, a0, i1, b0, j1, i0, j0);
      KLAUNCH(c, "k_ta_flag_commit", k_ta_flag_commit, dim3((j1 - b0 + 64) / 64, nz), dim3(64), d, s->dmv, s->limv, (const int *)s->dmk, b0, j1);
    }
  };

  int isv = is, iev = ie, jsv = js, jev = je, itt;
  for (itt = 1; itt <= max_iter; itt++) {
    if (isv > is - stencil) {   // :229-262
      std::vector<double *> f = { s->uhr, s->vhr, s->hprev };
      std::vector<int> stg = { 1, 2, 0 };
      for (int m = 0; m < ntr; m++) { f.push_back(Tr.t[m]); stg.push_back(0); }
      std::vector<int> nks(f.size(), nz);
      halo_wrap(c, f.data(), stg.data(), nks.data(), (int)f.size());
      const int nsten_halo = w / stencil;
      isv = is - nsten_halo * stencil; jsv = js - nsten_halo * stencil;
      iev = ie + nsten_halo * stencil; jev = je + nsten_halo * stencil;
      if ((nsten_halo > 1) || (itt == 1)) {
        KLAUNCH(c, "k_ta_flags_eval<0>", k_ta_flags_eval<0>, dim3(jev - jsv + 1, nz), dim3(64), d, (const double *)s->uhr, s->dmu,
                (const int *)s->dmk, jsv, jev, isv + stencil - 1, iev - stencil, (itt == 1) ? 1 : 0);
        KLAUNCH(c, "k_ta_flags_eval<1>", k_ta_flags_eval<1>, dim3(jev - jsv - 2 * stencil + 2, nz), dim3(64), d, (const double *)s->vhr, s->dmv,
                (const int *)s->dmk, jsv + stencil - 1, jev - stencil, isv + stencil, iev - stencil, (itt == 1) ? 1 : 0);
        KLAUNCH(c, "k_ta_dmk", k_ta_dmk, dim3(nz), dim3(64), d, (const int *)s->dmu, (const int *)s->dmv, s->dmk, jsv, jev,
                jsv + stencil - 1, jev - stencil);
      }
    }
    isv += stencil; iev -= stencil; jsv += stencil; jev -= stencil;
    if (x_first) {
      advect(0, isv, iev, jsv - stencil, jev + stencil);
      advect(1, isv, iev, jsv, jev);
      KLAUNCH(c, "k_ta_dmk", k_ta_dmk, dim3(nz), dim3(64), d, (const int *)s->dmu, (const int *)s->dmv, s->dmk, jsv - stencil,
              jev + stencil, jsv - 1, jev);
    } else {
      advect(1, isv - stencil, iev + stencil, jsv, jev);
      advect(0, isv, iev, jsv, jev);
      KLAUNCH(c, "k_ta_dmk", k_ta_dmk, dim3(nz), dim3(64), d, (const int *)s->dmu, (const int *)s->dmv, s->dmk, jsv, jev, jsv - 1, jev);
    }
    if (itt >= max_iter) break;
    if (isv > is - stencil) {   // sum_across_PEs(domore_k) :331
      { int rc_ = comm_allreduce_int_sum(c, s->dmk, nz); if (rc_) return rc_; }
      HIPCHK(hipMemcpyAsync(dmk_h.data(), s->dmk, nz * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      int do_any = 0;
      for (int k = 0; k < nz; k++) do_any += dmk_h[k];
      if (do_any == 0) break;
    }
  }
  if (rc_tile) return rc_tile;
  if (iters_out) *iters_out = itt;
  if (uhr_out) HIPCHK(hipMemcpyAsync(uhr_out, s->uhr, n3 * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (vhr_out) HIPCHK(hipMemcpyAsync(vhr_out, s->vhr, n3 * sizeof(double), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipGetLastError());
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  return MOM6X_OK;
}

// k_tridiag with the whole column on chip (the technique of k_vertvisc_cols, dyn_kernels.hip): c1 and the un-substituted T in
// registers, the un-substituted S of triDiagTS in LDS, one wavefront per work-group, inputs fetched TD_G layers ahead into a
// double buffer.  4 (5) words read and 1 (2) written per cell-layer instead of 9 (13); the same operations in the same order.
template <int NKT, bool TWO>   // (NKT: mom6x_dev.h NK_OF / NK_EXACT -- the layer count itself, or a bound on it)
__global__ void __launch_bounds__(64)
k_tridiag_cols(Dm d, const double *__restrict__ G, const double *__restrict__ hold, const double *__restrict__ ea,
               const double *__restrict__ eb, double *T, double *S, double h_neglect, int vertdiff,
               const double *__restrict__ sfc_flux, const double *__restrict__ btm_flux, double flux_scale, int i0, int i1, int j0,
               int j1) {
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;
  extern __shared__ double td_lds[];
  const int i = i0 + blockIdx.x * 64 + threadIdx.x;
  const int j = j0 + blockIdx.y;
  if (i > i1 || j > j1) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  double *ss = td_lds + threadIdx.x;
  double sfc_src = 0.0, btm_src = 0.0;
  if (vertdiff) {
    if (!(gm(G, d, MOM6X_G_mask2dT)[x] > 0.0)) return;
    if (sfc_flux) sfc_src = (flux_scale != 0.0) ? (sfc_flux[x] * flux_scale) : sfc_flux[x];
    if (btm_flux) btm_src = (flux_scale != 0.0) ? (btm_flux[x] * flux_scale) : btm_flux[x];
  }
  constexpr int TD_G = TWO ? 3 : 5, NG = (NK + TD_G - 1) / TD_G;
  double tt[NK], cc[NK];
  double q_h[2][TD_G], q_a[2][TD_G], q_b[2][TD_G], q_t[2][TD_G], q_s[2][TD_G];
  auto fetch = [&](int g, int b) {
#pragma unroll
    for (int m = 0; m < TD_G; m++) {
      const int k = g * TD_G + m;
      if (k < NK && k < nk) {
        const size_t c = x + (size_t)k * slab;
        q_h[b][m] = hold[c]; q_a[b][m] = ea[c]; q_b[b][m] = eb[c]; q_t[b][m] = T[c];
        if (TWO) q_s[b][m] = S[c];
      }
    }
  };
  double b1 = 0., d1 = 0., prev = 0., prevS = 0., eb_prev = 0.;
  fetch(0, 0);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) fetch(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < TD_G; m++) {
      const int k = g * TD_G + m;
      if (k < NK && k < nk) {
        const double h_tr = q_h[g & 1][m] + h_neglect, eak = q_a[g & 1][m], ebk = q_b[g & 1][m], Tk = q_t[g & 1][m];
        if (k == 0) {
          b1 = vertdiff ? 1.0 / ((h_tr + eak) + ebk) : 1.0 / (h_tr + ebk);
          d1 = h_tr * b1;
          prev = (b1 * h_tr) * Tk;
          if (vertdiff) prev = prev + b1 * sfc_src;
          if (TWO) prevS = (b1 * h_tr) * q_s[g & 1][m];
        } else {
          cc[k] = eb_prev * b1;
          const double b_denom_1 = h_tr + d1 * eak;
          b1 = 1.0 / (b_denom_1 + ebk);
          d1 = b_denom_1 * b1;
          if (vertdiff && k == nk - 1) prev = b1 * ((h_tr * Tk + btm_src) + eak * prev);
          else prev = b1 * (h_tr * Tk + eak * prev);
          if (TWO) prevS = b1 * (h_tr * q_s[g & 1][m] + eak * prevS);
        }
        eb_prev = ebk;
        tt[k] = prev;
        if (TWO) ss[k * 64] = prevS;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  T[x + (size_t)(nk - 1) * slab] = prev;
  if (TWO) S[x + (size_t)(nk - 1) * slab] = prevS;
  asm volatile("" ::: "memory");   // (S comes back from LDS, not from NK more live registers)
#pragma unroll
  for (int k = NK - 2; k >= 0; k--) {
    if (k >= nk - 1) continue;
    const size_t c = x + (size_t)k * slab;
    const double c1k = cc[k + 1];
    prev = tt[k] + c1k * prev;
    T[c] = prev;
    if (TWO) { prevS = ss[k * 64] + c1k * prevS; S[c] = prevS; }
    if (TWO && (k % 8) == 0) __builtin_amdgcn_sched_barrier(0);   // (or all 74 LDS reads are hoisted to the top: 150 registers)
  }
}

static int tridiag(mom6x_ctx *c, const double *hold, const double *ea, const double *eb, double *T, double *S, int vertdiff,
                   const double *sfc_flux, const double *btm_flux, double flux_scale, int is, int ie, int js, int je) {
  REQUIRE(c && hold && ea && eb && T, MOM6X_EINVAL, "tridiagonal solve: null array");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *c1;
  int rc;
  if ((rc = ctx_scratch(c, SCR_c1, d.nk, &c1))) return rc;
  static const bool walk = [] { const char *e = getenv("MOM6X_TRIDIAG"); return e && !strcmp(e, "walk"); }();
  if (d.nk <= COLS_NK_BOUND && !walk) {   // the layer counts the on-chip column kernel is built for
    const dim3 bc(64, 1, 1), gc((unsigned)((ie - is + 1 + 63) / 64), (unsigned)(je - js + 1), 1);
    // triDiagTS: T and S share the matrix (hold, ea, eb, and with them b1, d1, c1): ONE sweep for both, S's un-substituted values in
    // LDS -- 7 words per cell-layer instead of 10 (3.6 -> 3.1 ms per thermodynamic step at 1440 x 1080 x 75).  The instantiation with
    // a BOUND on the layer count serves 75 layers too: with the layer count itself the compiler schedules the fully unrolled,
    // branch-free column into 512 registers + 138 spilled; the uniform tests on the layer index keep its loads where they are.
#define TDC2(NKT) KLAUNCH_LDS(c, "k_tridiag_cols", (k_tridiag_cols<NKT, true>), gc, bc, (size_t)NK_OF(NKT) * 64 * sizeof(double), d, c->G, hold, ea, eb, T, S, \
                c->GV.H_subroundoff, vertdiff, sfc_flux, btm_flux, flux_scale, is, ie, js, je)
    if (S) {
      if (d.nk <= 52) TDC2(-52); else if (d.nk <= 66) TDC2(-66); else TDC2(-COLS_NK_BOUND);
      HIPCHK(hipGetLastError());
      return MOM6X_OK;
    }
#undef TDC2
#define TDC(NKT) KLAUNCH_LDS(c, "k_tridiag_cols", (k_tridiag_cols<NKT, false>), gc, bc, (size_t)0, d, c->G, hold, ea, eb, T, (double *)nullptr, \
                c->GV.H_subroundoff, vertdiff, sfc_flux, btm_flux, flux_scale, is, ie, js, je)
    COLS_NK_DISPATCH(d.nk, TDC);
#undef TDC
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  const dim3 b = blk2();
  KLAUNCH(c, "k_tridiag", k_tridiag, grid3(ie - is + 1, je - js + 1, 1, b), b, d, c->G, hold, ea, eb, T, S, c1, c->GV.H_subroundoff,
          vertdiff, sfc_flux, btm_flux, flux_scale, is, ie, js, je);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// triDiagTS(G, GV, is, ie, js, je, hold, ea, eb, T, S)  diabatic_aux.F90:394 (one field per call; S = second call)
extern "C" int mom6x_triDiagTS(mom6x_ctx *c, int is, int ie, int js, int je, const double *hold, const double *ea,
                               const double *eb, double *T, double *S) {
  return tridiag(c, hold, ea, eb, T, S, 0, nullptr, nullptr, 0.0, is, ie, js, je);
}
// triDiagTS_Eulerian(G, GV, is, ie, js, je, hold, ent, T, S)  :444 ; ent has nk+1 interfaces
extern "C" int mom6x_triDiagTS_Eulerian(mom6x_ctx *c, int is, int ie, int js, int je, const double *hold, const double *ent,
                                        double *T, double *S) {
  REQUIRE(ent, MOM6X_EINVAL, "triDiagTS_Eulerian: null ent");
  return mom6x_triDiagTS(c, is, ie, js, je, hold, ent, ent + c->dims.slab, T, S);
}
// tracer_vertdiff(h_old, ea, eb, dt, tr, G, GV, sfc_flux, btm_flux, ..., convert_flux_in)  tracer_diabatic.F90:25
extern "C" int mom6x_tracer_vertdiff(mom6x_ctx *c, const double *h_old, const double *ea, const double *eb, double dt,
                                     double *tr, const double *sfc_flux, const double *btm_flux, int convert_flux) {
  if (c && c->dims.nk == 1) return MOM6X_OK;   // the reference warns and returns
  const double scale = convert_flux ? dt * c->GV.RZ_to_H : 0.0;
  return tridiag(c, h_old, ea, eb, tr, nullptr, 1, sfc_flux, btm_flux, scale, 0, c->dims.ni - 1, 0, c->dims.nj - 1);
}
extern "C" int mom6x_tracer_vertdiff_Eulerian(mom6x_ctx *c, const double *h_old, const double *ent, double dt, double *tr,
                                              const double *sfc_flux, const double *btm_flux, int convert_flux) {
  REQUIRE(ent, MOM6X_EINVAL, "tracer_vertdiff_Eulerian: null ent");
  return mom6x_tracer_vertdiff(c, h_old, ent, ent + c->dims.slab, dt, tr, sfc_flux, btm_flux, convert_flux);
}
// tracer_vertdiff(..., btm_reservoir, sink_rate, ...) with sink_rate present (:123-179); btm_reservoir may be null
extern "C" int mom6x_tracer_vertdiff_sink(mom6x_ctx *c, const double *h_old, const double *ea, const double *eb, double dt, double *tr,
                                          const double *sfc_flux, const double *btm_flux, double *btm_reservoir, double sink_rate,
                                          int convert_flux) {
  REQUIRE(c && h_old && ea && eb && tr, MOM6X_EINVAL, "tracer_vertdiff: null array");
  if (c->dims.nk == 1) return MOM6X_OK;   // the reference warns and returns
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *c1, *snk, *hmd;
  int rc;
  if ((rc = ctx_scratch(c, SCR_c1, d.nk, &c1)) || (rc = ctx_scratch(c, SCR_t0, d.nk, &snk)) || (rc = ctx_scratch(c, SCR_t1, d.nk, &hmd))) return rc;
  const double scale = convert_flux ? dt * c->GV.RZ_to_H : 0.0;
  const double sink_dist = (dt * sink_rate) * c->GV.Z_to_H;
  const dim3 b = blk2();
  KLAUNCH(c, "k_tridiag_sink", k_tridiag_sink, grid3(d.ni, d.nj, 1, b), b, d, c->G, h_old, ea, eb, tr, c1, snk, hmd, c->GV.H_subroundoff,
          sfc_flux, btm_flux, scale, btm_reservoir, sink_dist, c->GV.H_to_RZ);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
extern "C" int mom6x_tracer_vertdiff_Eulerian_sink(mom6x_ctx *c, const double *h_old, const double *ent, double dt, double *tr,
                                                   const double *sfc_flux, const double *btm_flux, double *btm_reservoir,
                                                   double sink_rate, int convert_flux) {
  REQUIRE(c && ent, MOM6X_EINVAL, "tracer_vertdiff_Eulerian: null ent");
  return mom6x_tracer_vertdiff_sink(c, h_old, ent, ent + c->dims.slab, dt, tr, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux);
}
// diabatic(u, v, h, tv, BLD, fluxes, visc, ADp, CDp, dt, Time_end, G, GV, US, CS, ...)  diabatic_driver.F90:277
// Only the part of the dispatcher that is on the ported path: with GV%ke == 1 it returns immediately (:330),
// otherwise the host's mixing-coefficient physics (out of scope) supplies ea/eb or ent and calls the solvers above.
extern "C" int mom6x_diabatic_is_trivial(const mom6x_ctx *c) { return (c && c->dims.nk == 1) ? 1 : 0; }
