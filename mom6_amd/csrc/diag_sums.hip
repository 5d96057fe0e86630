// diag_sums.hip -- the order-invariant sums and checksums the reference's regression artefacts are made of, computed
// where the fields live:
//   reproducing_sum      MOM_coms.F90:235 (2-d), :349 (3-d), reproducing_EFP_sum_2d :96
//   chksum_{h,u,v,B}     MOM_checksums.F90:387 / :1413 (h), :1005 / :1782 (u), :1209 / :1986 (v), :688 / :1586 (B)
//   field_chksum         MOM_checksums.F90:2480 -> mpp_chksum (FMS): the restart files' `checksum` attribute
// One kernel reduces a rectangle of a pitched field, layer by layer, to integers: the six extended-fixed-point limbs of
// the reference's real_to_ints (prec = 2**46), the bit count of |x|, the wrapping sum of the bit patterns, min / max as
// order-preserving integer keys.  Integer addition commutes, so the block partials meet in atomics and the result is
// the reference's bit for bit whatever the order; the carries and the final regularize_ints run on the host on 6 x nk
// integers.  Across tiles the integers are summed by RCCL (sum_across_PEs of the reference).
#include <climits>
#include <vector>
#include "mom6x_dev.h"

namespace {

constexpr int NI_EFP = 6;
constexpr long long PREC = 1LL << 46;                                            // MOM_coms.F90:30
constexpr int MAX_COUNT_PREC = (1 << (63 - 46)) - 1;                             // :33
__host__ __device__ inline double efp_pr(int n) {                                // :40-41
  switch (n) { case 0: return 0x1p92; case 1: return 0x1p46; case 2: return 1.0; case 3: return 0x1p-46; case 4: return 0x1p-92; default: return 0x1p-138; }
}
__host__ __device__ inline double efp_I_pr(int n) {                              // :43-44
  switch (n) { case 0: return 0x1p-92; case 1: return 0x1p-46; case 2: return 1.0; case 3: return 0x1p46; case 4: return 0x1p92; default: return 0x1p138; }
}
constexpr double MAX_EFP_FLOAT = 0x1p92 * 0x1p63;                                // pr(1) * (2.**63 - 1.) :46

enum { M_SUM = 1, M_BC = 2, M_MINMAX = 4, M_CHK = 8 };
enum { F_NAN = 1, F_OVER = 2 };
// accumulators of one layer, structure of arrays over the layers: [SUMPART nk*8][MINPART nk][MAXPART nk*3]
enum { A_LIMB = 0, A_BC = 6, A_CHK = 7, A_NSUM = 8 };
struct AccPtr { long long *sum; long long *mn; long long *mx; };                 // mx: kmax, max |r| bits, flags
constexpr int RED_ROWS = 8;

__host__ __device__ inline long long order_key(double x) {                       // monotone map double -> int64
  long long b; memcpy(&b, &x, 8);
  return (b >= 0) ? b : (b ^ 0x7fffffffffffffffLL);
}
inline double key_value(long long k) {
  const long long b = (k >= 0) ? k : (k ^ 0x7fffffffffffffffLL);
  double x; memcpy(&x, &b, 8); return x;
}

__global__ void k_red_init(AccPtr A, int nk) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  for (int n = 0; n < A_NSUM; n++) A.sum[(size_t)k * A_NSUM + n] = 0;
  A.mn[k] = LLONG_MAX;
  A.mx[3 * k] = LLONG_MIN; A.mx[3 * k + 1] = 0; A.mx[3 * k + 2] = 0;
}

__device__ __forceinline__ long long wave_sum(long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}
__device__ __forceinline__ long long wave_min(long long v) {
  for (int o = 32; o > 0; o >>= 1) { const long long w = __shfl_down(v, o); v = (w < v) ? w : v; }
  return v;
}
__device__ __forceinline__ long long wave_max(long long v) {
  for (int o = 32; o > 0; o >>= 1) { const long long w = __shfl_down(v, o); v = (w > v) ? w : v; }
  return v;
}

// array: nk pitched planes; the rectangle (is..ie, js..je) in local compute indices; x = do_scale ? scale*a : a
__global__ void __launch_bounds__(256)
k_field_reduce(Dm d, const double *__restrict__ a, int is, int ie, int js, int je, int mode, int do_scale, double scale, AccPtr A) {
  const int i = is + blockIdx.x * 256 + threadIdx.x, k = blockIdx.z, jb = js + blockIdx.y * RED_ROWS;
  long long v[12];   // 0-5 limbs, 6 bit count, 7 pattern sum, 8 min key, 9 max key, 10 max |r| bits, 11 flags
  for (int n = 0; n < 8; n++) v[n] = 0;
  v[8] = LLONG_MAX; v[9] = LLONG_MIN; v[10] = 0; v[11] = 0;
  if (i <= ie) {
    for (int r = 0; r < RED_ROWS; r++) {
      const int j = jb + r;
      if (j > je) break;
      const double raw = a[ix3(d, i, j, k)];
      const double x = do_scale ? scale * raw : raw;
      if (mode & M_SUM) {                                                        // increment_ints_faster :652-682
        if ((x >= 1e30) == (x < 1e30)) v[11] |= F_NAN;
        else {
          double rs = fabs(x);
          long long rb; memcpy(&rb, &rs, 8);
          if (rb > v[10]) v[10] = rb;
          if (rs > MAX_EFP_FLOAT) v[11] |= F_OVER;
          else {
            const long long sgn = (x < 0.0) ? -1 : 1;
#pragma unroll
            for (int n = 0; n < NI_EFP; n++) {
              const long long ival = (long long)(rs * efp_I_pr(n));
              rs = rs - (double)ival * efp_pr(n);
              v[n] += sgn * ival;
            }
          }
        }
      }
      if (mode & M_BC) { const double ax = fabs(scale * raw); long long b; memcpy(&b, &ax, 8); v[6] += __popcll((unsigned long long)b); }
      if (mode & M_CHK) { long long b; memcpy(&b, &x, 8); v[7] = (long long)((unsigned long long)v[7] + (unsigned long long)b); }
      if (mode & M_MINMAX) {
        if (x != x) v[11] |= F_NAN;
        else { const long long key = order_key(x); v[8] = (key < v[8]) ? key : v[8]; v[9] = (key > v[9]) ? key : v[9]; }
      }
    }
  }
  __shared__ long long sm[4][12];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < 8; n++) v[n] = wave_sum(v[n]);
  v[8] = wave_min(v[8]); v[9] = wave_max(v[9]); v[10] = wave_max(v[10]);
  for (int o = 32; o > 0; o >>= 1) v[11] |= __shfl_down(v[11], o);
  if (lane == 0) for (int n = 0; n < 12; n++) sm[w][n] = v[n];
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int ww = 1; ww < 4; ww++) {
    for (int n = 0; n < 8; n++) v[n] += sm[ww][n];
    v[8] = (sm[ww][8] < v[8]) ? sm[ww][8] : v[8]; v[9] = (sm[ww][9] > v[9]) ? sm[ww][9] : v[9];
    v[10] = (sm[ww][10] > v[10]) ? sm[ww][10] : v[10]; v[11] |= sm[ww][11];
  }
  // exact carry of the block's limbs (each below 2048 * 2**46 here), so that any number of blocks can meet in int64
  for (int n = NI_EFP - 1; n >= 1; n--) { const long long c = v[n] / PREC; v[n] -= c * PREC; v[n - 1] += c; }
  long long *S = A.sum + (size_t)k * A_NSUM;
  if (mode & M_SUM) for (int n = 0; n < NI_EFP; n++) if (v[n]) atomicAdd((unsigned long long *)&S[n], (unsigned long long)v[n]);
  if (mode & M_BC) atomicAdd((unsigned long long *)&S[A_BC], (unsigned long long)v[6]);
  if (mode & M_CHK) atomicAdd((unsigned long long *)&S[A_CHK], (unsigned long long)v[7]);
  if (mode & M_MINMAX) { atomicMin(&A.mn[k], v[8]); atomicMax(&A.mx[3 * k], v[9]); }
  if (v[10]) atomicMax(&A.mx[3 * k + 1], v[10]);
  if (v[11]) atomicOr((unsigned long long *)&A.mx[3 * k + 2], (unsigned long long)v[11]);
}

// ---- host side of the extended fixed point type ---------------------------------------------------------------------
void carry_exact(long long *s) {   // carry_overflow :685 with exact integer division (the value is unchanged either way)
  for (int n = NI_EFP - 1; n >= 1; n--) if (llabs(s[n]) >= PREC) { const long long c = s[n] / PREC; s[n] -= c * PREC; s[n - 1] += c; }
}
void regularize_ints(long long *s) {                                             // :709-747
  carry_exact(s);
  bool positive = true;
  for (int n = 0; n < NI_EFP; n++) if (s[n] != 0) { if (s[n] < 0) positive = false; break; }
  if (positive) { for (int n = NI_EFP - 1; n >= 1; n--) if (s[n] < 0) { s[n] += PREC; s[n - 1] -= 1; } }
  else { for (int n = NI_EFP - 1; n >= 1; n--) if (s[n] > 0) { s[n] -= PREC; s[n - 1] += 1; } }
}
double ints_to_real(const long long *s) {                                        // :605-614
  double r = 0.0;
  for (int n = 0; n < NI_EFP; n++) r = r + efp_pr(n) * (double)s[n];
  return r;
}
bool increment_ints(long long *sum, const long long *b, long long lim) {         // :618-648; true: overflow_error
  for (int n = NI_EFP - 1; n >= 1; n--) {
    sum[n] += b[n];
    if (sum[n] > PREC) { sum[n] -= PREC; sum[n - 1] += 1; }
    else if (sum[n] < -PREC) { sum[n] += PREC; sum[n - 1] -= 1; }
  }
  sum[0] += b[0];
  return llabs(sum[0]) > lim;
}

struct HostAcc {
  std::vector<long long> sum, mn, mx;     // [nk*8], [nk], [nk*3]
  long long *limb(int k) { return &sum[(size_t)k * A_NSUM]; }
};

}  // namespace

int comm_allreduce_i64(mom6x_ctx *c, long long *dev, size_t n, int op);          // halo.hip: 0 min, 1 max, 2 sum
int comm_nranks(const mom6x_ctx *c);                                             // halo.hip

// Reduce the rectangle of `array` (nk planes) on the device, sum / min / max over the tiles, and bring the integers home.
static int reduce_run(mom6x_ctx *c, const double *array, int nk, int is, int ie, int js, int je, int mode, int do_scale,
                      double scale, bool across_PEs, HostAcc &H) {
  const mom6x_dims &D = c->dims;
  REQUIRE(array && nk >= 1, MOM6X_EINVAL, "reduce: null array or nk < 1");
  REQUIRE(is >= -D.halo - 1 && ie < D.ni + D.halo && js >= -D.halo - 1 && je < D.nj + D.halo && is <= ie + 1 && js <= je + 1,
          MOM6X_EINVAL, "reduce: the index range leaves the data domain");
  const size_t nwords = (size_t)nk * (A_NSUM + 1 + 3);
  if (c->red_cap < nwords) {
    if (c->red) HIPCHK(hipFree(c->red));
    c->red = nullptr; c->red_cap = 0;
    HIPCHK(hipMalloc(&c->red, nwords * sizeof(long long)));
    c->red_cap = nwords;
  }
  AccPtr A; A.sum = c->red; A.mn = A.sum + (size_t)nk * A_NSUM; A.mx = A.mn + nk;
  KLAUNCH(c, "k_red_init", k_red_init, dim3((nk + 63) / 64), dim3(64), A, nk);
  const int nx = ie - is + 1, ny = je - js + 1;
  if (nx > 0 && ny > 0)
    KLAUNCH(c, "k_field_reduce", k_field_reduce, dim3((nx + 255) / 256, (ny + RED_ROWS - 1) / RED_ROWS, nk), dim3(256), c->d,
            array, is, ie, js, je, mode, do_scale, scale, A);
  if (across_PEs && comm_nranks(c) > 1) {
    int rc;
    if ((rc = comm_allreduce_i64(c, A.sum, (size_t)nk * A_NSUM, 2)) || (rc = comm_allreduce_i64(c, A.mn, nk, 0)) ||
        (rc = comm_allreduce_i64(c, A.mx, (size_t)nk * 3, 1))) return rc;
  }
  H.sum.resize((size_t)nk * A_NSUM); H.mn.resize(nk); H.mx.resize((size_t)nk * 3);
  std::vector<long long> buf(nwords);
  HIPCHK(hipMemcpyAsync(buf.data(), c->red, nwords * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(H.sum.data(), buf.data(), H.sum.size() * 8);
  memcpy(H.mn.data(), buf.data() + H.sum.size(), H.mn.size() * 8);
  memcpy(H.mx.data(), buf.data() + H.sum.size() + H.mn.size(), H.mx.size() * 8);
  return MOM6X_OK;
}

// reproducing_sum_3d :349-558 (nk planes) / reproducing_sum_2d :235-343 (rank 2: one plane, `sums` must be null)
static int reproducing_sum_impl(mom6x_ctx *c, const double *array, int nk, int rank, int is, int ie, int js, int je, double unscale,
                                int only_on_PE, double *sum, double *sums, int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err) {
  REQUIRE(c && sum, MOM6X_EINVAL, "mom6x_reproducing_sum: null argument");
  REQUIRE(rank == 3 || (nk == 1 && !sums && !EFP_lay_sums), MOM6X_EINVAL, "mom6x_reproducing_sum: a 2-d sum has one plane and no layer sums");
  const int np = only_on_PE ? 1 : comm_nranks(c);
  REQUIRE(np <= MAX_COUNT_PREC, MOM6X_EINVAL,
          "reproducing_sum: Too many processors are being used for the value of prec.  Reduce prec to (2^63-1)/num_PEs.");
  const long long prec_error = ((1LL << 62) + ((1LL << 62) - 1)) / np;
  const int do_unscale = (unscale != 1.0);
  HostAcc H;
  int rc = reduce_run(c, array, nk, is, ie, js, je, M_SUM, do_unscale, unscale, !only_on_PE, H);
  if (rc) return rc;
  bool nan = false, over = false; double max_mag = 0.0;
  for (int k = 0; k < nk; k++) {
    const long long f = H.mx[3 * k + 2];
    nan |= (f & F_NAN) != 0; over |= (f & F_OVER) != 0;
    double m; memcpy(&m, &H.mx[3 * k + 1], 8);
    if (m > max_mag) max_mag = m;
  }
  const bool conv_over = (max_mag >= (double)prec_error * efp_pr(0));
  const bool lay = (sums || EFP_lay_sums);
  std::vector<long long> tot(NI_EFP, 0);
  if (!lay) for (int k = 0; k < nk; k++) {      // one running sum over all the layers, carried as it goes
    for (int n = 0; n < NI_EFP; n++) tot[n] += H.limb(k)[n];
    carry_exact(tot.data());
  }
  for (int k = 0; k < nk; k++) { carry_exact(H.limb(k)); if (llabs(H.limb(k)[0]) > prec_error) over = true; }
  if (llabs(tot[0]) > prec_error) over = true;
  if (err) {
    *err = 0;
    if (rank == 3) { if (conv_over) *err += 1; if (over) *err += 2; if (nan) *err += 2; }   // :480-483, :524-527
    else { if (over) *err += 2; if (nan) *err += 4; }                                        // :205-209
    if (*err > 0) { for (int k = 0; k < nk; k++) for (int n = 0; n < NI_EFP; n++) H.limb(k)[n] = 0; for (auto &t : tot) t = 0; }
  } else {
    REQUIRE(!nan, MOM6X_ENUMERIC, rank == 3 ? "NaN in input field of reproducing_sum(_3d)." : "NaN in input field of reproducing_EFP_sum(_2d).");
    if (conv_over) {
      mom6x_set_error("Overflow in reproducing_%s conversion of %13.5E", rank == 3 ? "sum(_3d)" : "EFP_sum(_2d)", max_mag);
      return MOM6X_ENUMERIC;
    }
    REQUIRE(!over, MOM6X_ENUMERIC, rank == 3 ? "Overflow in reproducing_sum(_3d)." : "Overflow in reproducing_EFP_sum(_2d).");
  }
  double s = 0.0;
  if (lay) {
    for (int k = 0; k < nk; k++) {
      regularize_ints(H.limb(k));
      const double val = ints_to_real(H.limb(k));
      if (sums) sums[k] = val;
      s = s + val;
    }
    if (EFP_lay_sums) for (int k = 0; k < nk; k++) for (int n = 0; n < NI_EFP; n++) EFP_lay_sums[(size_t)k * NI_EFP + n] = H.limb(k)[n];
    if (EFP_sum) {
      long long e[NI_EFP] = {0, 0, 0, 0, 0, 0};
      for (int k = 0; k < nk; k++) increment_ints(e, H.limb(k), PREC);
      for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = e[n];
    }
  } else {
    regularize_ints(tot.data());
    s = ints_to_real(tot.data());
    if (EFP_sum) for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = tot[n];
  }
  if (rank == 3) {
    if (do_unscale) {
      const double I_unscale = (fabs(unscale) > 0.0) ? 1.0 / unscale : 0.0;
      s = s * I_unscale;
      if (sums) for (int k = 0; k < nk; k++) sums[k] = sums[k] * I_unscale;
    }
  } else {
    double I_unscale = 1.0;
    if (do_unscale && fabs(unscale) > 0.0) I_unscale = 1.0 / unscale;
    s = s * I_unscale;
  }
  *sum = s;
  return MOM6X_OK;
}

extern "C" int mom6x_reproducing_sum_3d(mom6x_ctx *c, const double *array, int nk, int is, int ie, int js, int je, double unscale,
                                        int only_on_PE, double *sum, double *sums, int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err) {
  return reproducing_sum_impl(c, array, nk, 3, is, ie, js, je, unscale, only_on_PE, sum, sums, EFP_sum, EFP_lay_sums, err);
}
extern "C" int mom6x_reproducing_sum_2d(mom6x_ctx *c, const double *array, int is, int ie, int js, int je, double unscale, int only_on_PE,
                                        double *sum, int64_t *EFP_sum, int *err) {
  return reproducing_sum_impl(c, array, 1, 2, is, ie, js, je, unscale, only_on_PE, sum, nullptr, EFP_sum, nullptr, err);
}

// subchk of the chksum_* routines: the h-point computational domain shifted by (di, dj); a default integer that wraps
static int subchk(mom6x_ctx *c, const double *array, int nk, int di, int dj, double scaling, int *bc) {
  HostAcc H;
  const int rc = reduce_run(c, array, nk, di, c->dims.ni - 1 + di, dj, c->dims.nj - 1 + dj, M_BC, 1, scaling, true, H);
  if (rc) return rc;
  unsigned long long s = 0;
  for (int k = 0; k < nk; k++) s += (unsigned long long)H.limb(k)[A_BC];
  *bc = (int)((int32_t)(uint32_t)s % 1000000000);                                // bc_modulus :110; mod keeps the dividend's sign
  return MOM6X_OK;
}

extern "C" int mom6x_chksum(mom6x_ctx *c, const double *array, int nk, int rank, int stagger, int haloshift, int symmetric,
                            int omit_corners, const double *scale, mom6x_chksum_result *out) {
  REQUIRE(c && array && out, MOM6X_EINVAL, "mom6x_chksum: null argument");
  REQUIRE(stagger >= 0 && stagger <= 3 && (rank == 2 || rank == 3) && nk >= 1 && (rank == 3 || nk == 1), MOM6X_EINVAL,
          "mom6x_chksum: bad stagger / rank / nk");
  const mom6x_dims &D = c->dims;
  const int ni = D.ni, nj = D.nj;
  int hs = haloshift;
  if (hs < 0) hs = D.halo;                                                       // hshift = HI%ied-HI%iec :1499
  REQUIRE(hs <= D.halo, MOM6X_EINVAL, "Error in chksum: haloshift is wider than the halo");
  const double scaling = scale ? *scale : 1.0;
  HostAcc H;
  int rc;
  // checkForNaNs :1453 over the h-point domain, and the mean of subStats = reproducing_sum(h-point domain) / n
  double mean;
  rc = reduce_run(c, array, nk, 0, ni - 1, 0, nj - 1, M_SUM | M_MINMAX, scale != nullptr, scaling, true, H);
  if (rc) return rc;
  for (int k = 0; k < nk; k++) REQUIRE(!(H.mx[3 * k + 2] & F_NAN), MOM6X_ENUMERIC, "NaN detected in chksum");
  {
    std::vector<long long> tot(NI_EFP, 0);
    for (int k = 0; k < nk; k++) { for (int n = 0; n < NI_EFP; n++) tot[n] += H.limb(k)[n]; carry_exact(tot.data()); }
    for (int k = 0; k < nk; k++) REQUIRE(!(H.mx[3 * k + 2] & F_OVER), MOM6X_ENUMERIC, "Overflow in reproducing_sum(_3d).");
    regularize_ints(tot.data());
    mean = ints_to_real(tot.data());
  }
  const long long npts = (long long)ni * nj * nk * comm_nranks(c);
  out->mean = mean / (double)npts;
  // min / max over the staggered computational domain (symmetric: one more row / column to the west / south)
  int sym_stats = symmetric;
  if (stagger != 0 && haloshift > 0) sym_stats = 1;
  const int Is = ((stagger == 1 || stagger == 3) && sym_stats) ? -1 : 0, Js = ((stagger == 2 || stagger == 3) && sym_stats) ? -1 : 0;
  if (Is != 0 || Js != 0) {
    rc = reduce_run(c, array, nk, Is, ni - 1, Js, nj - 1, M_MINMAX, scale != nullptr, scaling, true, H);
    if (rc) return rc;
  }
  long long kmin = LLONG_MAX, kmax = LLONG_MIN;
  for (int k = 0; k < nk; k++) { if (H.mn[k] < kmin) kmin = H.mn[k]; if (H.mx[3 * k] > kmax) kmax = H.mx[3 * k]; }
  out->amin = 0. + key_value(kmin); out->amax = 0. + key_value(kmax);            // as chk_sum_msg3 :2638 prints them
  // the bit counts
  for (int n = 0; n < 4; n++) out->bc[n] = 0;
  out->nbc = 0; out->bc_kind = MOM6X_CHK_NONE;
  if ((rc = subchk(c, array, nk, 0, 0, scaling, &out->bc0))) return rc;
  int sh[4][2]; int n = 0, kind = MOM6X_CHK_NONE;
#define SH(a, b) do { sh[n][0] = (a); sh[n][1] = (b); n++; } while (0)
  if (stagger == 0) {
    if (hs != 0) {
      if (!omit_corners) { SH(-hs, -hs); SH(hs, -hs); SH(-hs, hs); SH(hs, hs); kind = MOM6X_CHK_CORNERS; }
      else { SH(0, hs); SH(0, -hs); SH(hs, 0); SH(-hs, 0); kind = MOM6X_CHK_NSEW; }
    }
  } else if (!(hs == 0 && !symmetric)) {
    const int sym = symmetric ? 1 : 0;
    if (stagger == 1 || stagger == 2) {
      const int wx = (stagger == 1) ? sym : 0, wy = (stagger == 2) ? sym : 0;
      if (hs == 0) { if (stagger == 1) { SH(-1, 0); kind = MOM6X_CHK_W; } else { SH(0, -1); kind = MOM6X_CHK_S; } }
      else if (!omit_corners) { SH(-hs - wx, -hs - wy); SH(hs, -hs - wy); SH(-hs - wx, hs); SH(hs, hs); kind = MOM6X_CHK_CORNERS; }
      else { SH(0, hs); SH(0, -hs - wy); SH(hs, 0); SH(-hs - wx, 0); kind = MOM6X_CHK_NSEW; }
    } else if (rank == 2) {                                                      // chksum_B_2d :767-788
      if (!omit_corners) { SH(-hs - sym, -hs - sym); SH(hs, -hs - sym); SH(-hs - sym, hs); SH(hs, hs); kind = MOM6X_CHK_CORNERS; }
      else { SH(0, hs); SH(0, -hs); SH(hs, 0); SH(-hs, 0); kind = MOM6X_CHK_NSEW; }
    } else {                                                                     // chksum_B_3d :1665-1690
      if (!omit_corners) { SH(-hs - 1, -hs - 1); SH(hs, -hs - 1); SH(-hs - 1, hs); SH(hs, hs); kind = MOM6X_CHK_CORNERS; }
      else { SH(0, hs); SH(0, -hs - sym); SH(hs, 0); SH(-hs - sym, 0); kind = MOM6X_CHK_NSEW; }
    }
  }
#undef SH
  for (int m = 0; m < n; m++) {
    REQUIRE(sh[m][0] >= -D.halo - 1 && sh[m][0] <= D.halo && sh[m][1] >= -D.halo - 1 && sh[m][1] <= D.halo, MOM6X_EINVAL,
            "Error in chksum: the shifted domain leaves the data domain");
    if ((rc = subchk(c, array, nk, sh[m][0], sh[m][1], scaling, &out->bc[m]))) return rc;
  }
  out->nbc = n; out->bc_kind = kind;
  return MOM6X_OK;
}

extern "C" int mom6x_field_chksum(mom6x_ctx *c, const double *array, int nk, int is, int ie, int js, int je, double unscale,
                                  int64_t *chksum) {
  REQUIRE(c && chksum, MOM6X_EINVAL, "mom6x_field_chksum: null argument");
  HostAcc H;
  const int rc = reduce_run(c, array, nk, is, ie, js, je, M_CHK, unscale != 1.0, unscale, true, H);
  if (rc) return rc;
  unsigned long long s = 0;
  for (int k = 0; k < nk; k++) s += (unsigned long long)H.limb(k)[A_CHK];
  *chksum = (int64_t)s;
  return MOM6X_OK;
}

void diag_sums_free(mom6x_ctx *c) { if (c->red) (void)hipFree(c->red); c->red = nullptr; c->red_cap = 0; }
