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
  A.mx[4 * k] = LLONG_MIN; A.mx[4 * k + 1] = 0; A.mx[4 * k + 2] = 0; A.mx[4 * k + 3] = 0;
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

// What is reduced: a field as it is, or one of the integrands of write_energy (MOM_sum_output.F90) formed on the fly.
struct SrcArray {            // nk pitched planes
  const double *a;
  __device__ __forceinline__ double operator()(const Dm &d, int i, int j, int k) const { return a[ix3(d, i, j, k)]; }
};
struct SrcMass {             // tmp1 = h * (H_to_RZ*areaTm) :506-508
  const double *h, *G; double H_to_RZ;
  __device__ __forceinline__ double operator()(const Dm &d, int i, int j, int k) const {
    const size_t c = ix2(d, i, j);
    const double areaTm = gm(G, d, MOM6X_G_mask2dT)[c] * gm(G, d, MOM6X_G_areaT)[c];
    return h[c + (size_t)k * d.slab] * (H_to_RZ * areaTm);
  }
};
struct SrcKE {               // tmp1 = (0.25*H_to_RZ*(areaTm*h)) * (((u(I-1)**2) + (u(I)**2)) + ((v(J-1)**2) + (v(J)**2))) :669-673
  const double *h, *u, *v, *G; double H_to_RZ;
  __device__ __forceinline__ double operator()(const Dm &d, int i, int j, int k) const {
    const size_t c = ix2(d, i, j), c3 = c + (size_t)k * d.slab;
    const double areaTm = gm(G, d, MOM6X_G_mask2dT)[c] * gm(G, d, MOM6X_G_areaT)[c];
    const double uw = u[c3 - 1], ue = u[c3], vs = v[c3 - d.pitch], vn = v[c3];
    return (0.25 * H_to_RZ * (areaTm * h[c3])) * (((uw * uw) + (ue * ue)) + ((vs * vs) + (vn * vn)));
  }
};

// The rectangle (is..ie, js..je) in local compute indices of nk layers; x = do_scale ? scale*src : src
template <class Src>
__global__ void __launch_bounds__(256)
k_field_reduce(Dm d, Src src, int is, int ie, int js, int je, int mode, int do_scale, double scale, AccPtr A) {
  const int i = is + blockIdx.x * 256 + threadIdx.x, k = blockIdx.z, jb = js + blockIdx.y * RED_ROWS;
  long long v[12];   // 0-5 limbs, 6 bit count, 7 pattern sum, 8 min key, 9 max key, 10 max |r| bits, 11 flags
  for (int n = 0; n < 8; n++) v[n] = 0;
  v[8] = LLONG_MAX; v[9] = LLONG_MIN; v[10] = 0; v[11] = 0;
  if (i <= ie) {
    for (int r = 0; r < RED_ROWS; r++) {
      const int j = jb + r;
      if (j > je) break;
      const double raw = src(d, i, j, k);
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
  if (mode & M_MINMAX) { atomicMin(&A.mn[k], v[8]); atomicMax(&A.mx[4 * k], v[9]); }
  if (v[10]) atomicMax(&A.mx[4 * k + 1], v[10]);
  // one 0/1 word per flag: the tiles' words meet in a MAX all-reduce, which is an OR only for single bits
  if (v[11] & F_NAN) atomicMax(&A.mx[4 * k + 2], 1LL);
  if (v[11] & F_OVER) atomicMax(&A.mx[4 * k + 3], 1LL);
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
template <class Src>
static int reduce_src(mom6x_ctx *c, const Src &src, const char *kname, int nk, int is, int ie, int js, int je, int mode, int do_scale,
                      double scale, bool across_PEs, HostAcc &H) {
  const mom6x_dims &D = c->dims;
  REQUIRE(nk >= 1, MOM6X_EINVAL, "reduce: nk < 1");
  REQUIRE(is >= -D.halo - 1 && ie < D.ni + D.halo && js >= -D.halo - 1 && je < D.nj + D.halo && is <= ie + 1 && js <= je + 1,
          MOM6X_EINVAL, "reduce: the index range leaves the data domain");
  const size_t nwords = (size_t)nk * (A_NSUM + 1 + 4);
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
    KLAUNCH(c, kname, k_field_reduce<Src>, dim3((nx + 255) / 256, (ny + RED_ROWS - 1) / RED_ROWS, nk), dim3(256), c->d,
            src, is, ie, js, je, mode, do_scale, scale, A);
  if (across_PEs && comm_nranks(c) > 1) {
    int rc;
    if ((rc = comm_allreduce_i64(c, A.sum, (size_t)nk * A_NSUM, 2)) || (rc = comm_allreduce_i64(c, A.mn, nk, 0)) ||
        (rc = comm_allreduce_i64(c, A.mx, (size_t)nk * 4, 1))) return rc;
  }
  H.sum.resize((size_t)nk * A_NSUM); H.mn.resize(nk); H.mx.resize((size_t)nk * 4);
  std::vector<long long> buf(nwords);
  HIPCHK(hipMemcpyAsync(buf.data(), c->red, nwords * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(H.sum.data(), buf.data(), H.sum.size() * 8);
  memcpy(H.mn.data(), buf.data() + H.sum.size(), H.mn.size() * 8);
  memcpy(H.mx.data(), buf.data() + H.sum.size() + H.mn.size(), H.mx.size() * 8);
  return MOM6X_OK;
}

static int reduce_run(mom6x_ctx *c, const double *array, int nk, int is, int ie, int js, int je, int mode, int do_scale,
                      double scale, bool across_PEs, HostAcc &H) {
  REQUIRE(array, MOM6X_EINVAL, "reduce: null array");
  SrcArray src; src.a = array;
  return reduce_src(c, src, "k_field_reduce", nk, is, ie, js, je, mode, do_scale, scale, across_PEs, H);
}

// reproducing_sum_3d :349-558 (nk planes) / reproducing_sum_2d :235-343 (rank 2: one plane, `sums` must be null).
// The integers come from `H` (already reduced); everything after the loops of the reference happens here.
static int reproducing_sum_finish(mom6x_ctx *c, HostAcc &H, int nk, int rank, double unscale, int only_on_PE, double *sum, double *sums,
                                  int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err);

static int reproducing_sum_impl(mom6x_ctx *c, const double *array, int nk, int rank, int is, int ie, int js, int je, double unscale,
                                int only_on_PE, double *sum, double *sums, int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err) {
  REQUIRE(c && sum, MOM6X_EINVAL, "mom6x_reproducing_sum: null argument");
  REQUIRE(rank == 3 || (nk == 1 && !sums && !EFP_lay_sums), MOM6X_EINVAL, "mom6x_reproducing_sum: a 2-d sum has one plane and no layer sums");
  HostAcc H;
  int rc = reduce_run(c, array, nk, is, ie, js, je, M_SUM, unscale != 1.0, unscale, !only_on_PE, H);
  if (rc) return rc;
  return reproducing_sum_finish(c, H, nk, rank, unscale, only_on_PE, sum, sums, EFP_sum, EFP_lay_sums, err);
}

static int reproducing_sum_finish(mom6x_ctx *c, HostAcc &H, int nk, int rank, double unscale, int only_on_PE, double *sum, double *sums,
                                  int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err) {
  const int np = only_on_PE ? 1 : comm_nranks(c);
  REQUIRE(np <= MAX_COUNT_PREC, MOM6X_EINVAL,
          "reproducing_sum: Too many processors are being used for the value of prec.  Reduce prec to (2^63-1)/num_PEs.");
  const long long prec_error = ((1LL << 62) + ((1LL << 62) - 1)) / np;
  const int do_unscale = (unscale != 1.0);
  bool nan = false, over = false; double max_mag = 0.0;
  for (int k = 0; k < nk; k++) {
    nan |= H.mx[4 * k + 2] != 0; over |= H.mx[4 * k + 3] != 0;
    double m; memcpy(&m, &H.mx[4 * k + 1], 8);
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
  for (int k = 0; k < nk; k++) REQUIRE(!H.mx[4 * k + 2], MOM6X_ENUMERIC, "NaN detected in chksum");
  {
    std::vector<long long> tot(NI_EFP, 0);
    for (int k = 0; k < nk; k++) { for (int n = 0; n < NI_EFP; n++) tot[n] += H.limb(k)[n]; carry_exact(tot.data()); }
    for (int k = 0; k < nk; k++) REQUIRE(!H.mx[4 * k + 3], MOM6X_ENUMERIC, "Overflow in reproducing_sum(_3d).");
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
  for (int k = 0; k < nk; k++) { if (H.mn[k] < kmin) kmin = H.mn[k]; if (H.mx[4 * k] > kmax) kmax = H.mx[4 * k]; }
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


// ---- write_energy (MOM_sum_output.F90:321) ---------------------------------------------------------------------------
namespace {

struct DiagState {                          // the members of Sum_output_CS the sums of write_energy read
  mom6x_sum_output_params p;
  int listsize;                             // Depth_List: 1-based arrays of listsize entries
  std::vector<double> depth, area, vol_below;
  std::vector<int> lH;                      // CS%lH: where each interface's volume was found last time
  std::vector<double> g_prime;
  double *g_prime_dev, *z0_dev;             // GV%g_prime(1:nk), Z_0APE(1:nk+1)
  long long *cfl_dev;                       // the two CFL maxima as bit patterns (non-negative doubles order like integers)
};

// PE_pt of the Boussinesq branch :623-634: one thread per column, from the bottom up; plane nk+1 is zero
__global__ void __launch_bounds__(256)
k_pe_pt(Dm d, const double *__restrict__ G, const double *__restrict__ h, const double *__restrict__ Z_0APE,
        const double *__restrict__ g_prime, double Rho0, double H_to_Z, double Z_ref, double *__restrict__ PE_pt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= d.ni) return;
  const size_t c = ix2(d, i, j);
  const double areaTm = gm(G, d, MOM6X_G_mask2dT)[c] * gm(G, d, MOM6X_G_areaT)[c];
  const double Dref = gm(G, d, MOM6X_G_bathyT)[c] + Z_ref;
  double hbelow = 0.0;
  PE_pt[c + (size_t)d.nk * d.slab] = 0.0;
  for (int K = d.nk - 1; K >= 0; K--) {
    hbelow = hbelow + h[c + (size_t)K * d.slab] * H_to_Z;
    const double hint = Z_0APE[K] + (hbelow - Dref);
    double hbot = Z_0APE[K] - Dref;
    hbot = (hbot + fabs(hbot)) * 0.5;
    PE_pt[c + (size_t)K * d.slab] = (0.5 * areaTm) * (Rho0 * g_prime[K]) * (hint * hint - hbot * hbot);
  }
}

// Salt_int, Temp_int :677-682: the salt and heat content of every column, layers added from the top down
__global__ void __launch_bounds__(256)
k_ts_int(Dm d, const double *__restrict__ G, const double *__restrict__ h, const double *__restrict__ T, const double *__restrict__ S,
         double C_p, double H_to_RZ, double *__restrict__ Salt_int, double *__restrict__ Temp_int) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= d.ni) return;
  const size_t c = ix2(d, i, j);
  const double areaTm = gm(G, d, MOM6X_G_mask2dT)[c] * gm(G, d, MOM6X_G_areaT)[c];
  double si = 0.0, ti = 0.0;
  for (int k = 0; k < d.nk; k++) {
    const size_t c3 = c + (size_t)k * d.slab;
    const double hm = h[c3] * (H_to_RZ * areaTm);
    si = si + S[c3] * hm;
    ti = ti + (C_p * T[c3]) * hm;
  }
  Salt_int[c] = si; Temp_int[c] = ti;
}

// max_CFL :701-722: u faces I = -1..ni-1 of rows 0..nj-1, v faces J = -1..nj-1 of columns 0..ni-1.  One thread per column
// walks all layers with the five metric values of its two faces in registers (the planes are read once, not once per layer).
__global__ void __launch_bounds__(256)
k_max_cfl(Dm d, const double *__restrict__ G, const double *__restrict__ u, const double *__restrict__ v, double dt,
          long long *__restrict__ out) {
  const int i = -1 + blockIdx.x * 256 + threadIdx.x, j = -1 + blockIdx.y;
  double m1 = 0.0, m2 = 0.0;
  if (i < d.ni) {
    const size_t c = ix2(d, i, j);
    const double *IareaT = gm(G, d, MOM6X_G_IareaT);
    const bool do_u = (j >= 0), do_v = (i >= 0);
    const double Ia_c = IareaT[c], Ia_e = IareaT[c + 1], Ia_n = IareaT[c + d.pitch];
    const double dyCu = gm(G, d, MOM6X_G_dy_Cu)[c], IdxCu = gm(G, d, MOM6X_G_IdxCu)[c];
    const double dxCv = gm(G, d, MOM6X_G_dx_Cv)[c], IdyCv = gm(G, d, MOM6X_G_IdyCv)[c];
    for (int k = 0; k < d.nk; k++) {
      const size_t c3 = c + (size_t)k * d.slab;
      if (do_u) {
        const double uu = u[c3];
        double CFL_Iarea = Ia_c;
        if (uu < 0.0) CFL_Iarea = Ia_e;
        const double CFL_trans = fabs(uu * dt) * (dyCu * CFL_Iarea);
        const double CFL_lin = fabs(uu * dt) * IdxCu;
        if (CFL_trans > m1) m1 = CFL_trans;          // (a NaN never replaces the running maximum, as in max(max_CFL, NaN))
        if (CFL_lin > m2) m2 = CFL_lin;
      }
      if (do_v) {
        const double vv = v[c3];
        double CFL_Iarea = Ia_c;
        if (vv < 0.0) CFL_Iarea = Ia_n;
        const double CFL_trans = fabs(vv * dt) * (dxCv * CFL_Iarea);
        const double CFL_lin = fabs(vv * dt) * IdyCv;
        if (CFL_trans > m1) m1 = CFL_trans;
        if (CFL_lin > m2) m2 = CFL_lin;
      }
    }
  }
  long long b1, b2;
  memcpy(&b1, &m1, 8); memcpy(&b2, &m2, 8);
  b1 = wave_max(b1); b2 = wave_max(b2);
  // one atomic per wavefront at most, and only when it can still raise the running maximum (a stale read is harmless)
  if ((threadIdx.x & 63) == 0) {
    if (b1 > __atomic_load_n(&out[0], __ATOMIC_RELAXED)) atomicMax(&out[0], b1);
    if (b2 > __atomic_load_n(&out[1], __ATOMIC_RELAXED)) atomicMax(&out[1], b2);
  }
}

void diag_state_free(DiagState *s) {
  if (!s) return;
  (void)hipFree(s->g_prime_dev); (void)hipFree(s->z0_dev); (void)hipFree(s->cfl_dev);
  delete s;
}

}  // namespace

int comm_allreduce_f64(mom6x_ctx *c, double *dev, size_t n, int op);             // halo.hip

// create_depth_list :1203-1326.  Dlist / AreaList hold the GLOBAL lists (entries of other tiles arrive by the sum).
static void create_depth_list(DiagState *S, const std::vector<double> &Dlist, const std::vector<double> &AreaList, int mls) {
  const double min_depth_inc = S->p.D_list_min_inc;
  std::vector<int> indx2((size_t)mls + 2);
  for (int j = 1; j <= mls + 1; j++) indx2[j] = j;
  int k = mls / 2 + 1, ir = mls, indxt, i, j;
  double Dnow;
  for (;;) {                                                                     // the heap sort of :1246-1267
    if (k > 1) { k = k - 1; indxt = indx2[k]; Dnow = Dlist[indxt]; }
    else {
      indxt = indx2[ir]; Dnow = Dlist[indxt];
      indx2[ir] = indx2[1];
      ir = ir - 1;
      if (ir == 1) { indx2[1] = indxt; break; }
    }
    i = k; j = k * 2;
    for (;;) {
      if (j > ir) break;
      if (j < ir && Dlist[indx2[j]] < Dlist[indx2[j + 1]]) j = j + 1;
      if (Dnow < Dlist[indx2[j]]) { indx2[i] = indx2[j]; i = j; j = j + i; }
      else j = ir + 1;
    }
    indx2[i] = indxt;
  }
  double D_list_prev = Dlist[indx2[mls]];
  int list_size = 2;
  for (k = mls - 1; k >= 1; k--) if (Dlist[indx2[k]] < D_list_prev - min_depth_inc) { list_size++; D_list_prev = Dlist[indx2[k]]; }
  const int listsize = list_size + 1;
  S->listsize = listsize;
  S->depth.assign((size_t)listsize + 1, 0.0); S->area.assign((size_t)listsize + 1, 0.0); S->vol_below.assign((size_t)listsize + 1, 0.0);
  double vol = 0.0, area = 0.0, Dprev = Dlist[indx2[mls]];
  D_list_prev = Dprev;
  int kl = 0;
  for (k = mls; k >= 1; k--) {
    i = indx2[k];
    vol = vol + area * (Dprev - Dlist[i]);
    area = area + AreaList[i];
    bool add_to_list = false;
    if (kl == 0 || k == 1) add_to_list = true;
    else if (Dlist[indx2[k - 1]] < D_list_prev - min_depth_inc) { add_to_list = true; D_list_prev = Dlist[indx2[k - 1]]; }
    if (add_to_list) { kl++; S->depth[kl] = Dlist[i]; S->area[kl] = area; S->vol_below[kl] = vol; }
    Dprev = Dlist[i];
  }
  while (kl + 1 < listsize) {
    kl++;
    S->vol_below[kl] = S->vol_below[kl - 1] * 1.000001; S->area[kl] = S->area[kl - 1]; S->depth[kl] = S->depth[kl - 1];
  }
  S->vol_below[listsize] = S->vol_below[listsize - 1] * 1000.0;
  S->area[listsize] = S->area[listsize - 1];
  S->depth[listsize] = S->depth[listsize - 1];
}

// MOM_sum_output_init :147 (the members the sums need) + depth_list_setup :1161 (the list is created, not read)
extern "C" int mom6x_sum_output_init(mom6x_ctx *c, const mom6x_sum_output_params *p, const double *g_prime) {
  REQUIRE(c && p && g_prime, MOM6X_EINVAL, "mom6x_sum_output_init: null argument");
  REQUIRE(c->GV.Boussinesq == 1, MOM6X_EUNSUPPORTED, "write_energy: only the Boussinesq branches are on the device path");
  const mom6x_dims &D = c->dims;
  const int nk = D.nk;
  diag_state_free((DiagState *)c->diag); c->diag = nullptr;
  DiagState *S = new DiagState();
  S->p = *p; S->g_prime_dev = S->z0_dev = nullptr; S->cfl_dev = nullptr; S->listsize = 0;
  c->diag = S;
  S->g_prime.assign(g_prime, g_prime + nk);
  HIPCHK(hipMalloc(&S->g_prime_dev, sizeof(double) * nk));
  HIPCHK(hipMalloc(&S->z0_dev, sizeof(double) * (nk + 1)));
  HIPCHK(hipMalloc(&S->cfl_dev, sizeof(long long) * 2));
  HIPCHK(hipMemcpy(S->g_prime_dev, g_prime, sizeof(double) * nk, hipMemcpyHostToDevice));
  if (p->do_APE_calc) {
    // Dlist, AreaList :1226-1237: this tile's entries of the global lists, the others by sum_across_PEs
    const int mls = D.ni_glob * D.nj_glob;
    std::vector<double> plane((size_t)D.slab * 3);
    HIPCHK(hipStreamSynchronize(c->stream));
    const int ids[3] = {MOM6X_G_bathyT, MOM6X_G_mask2dT, MOM6X_G_areaT};
    for (int m = 0; m < 3; m++)
      HIPCHK(hipMemcpy(plane.data() + (size_t)m * D.slab, c->G + (size_t)ids[m] * D.slab, sizeof(double) * D.slab, hipMemcpyDeviceToHost));
    std::vector<double> lists(2 * ((size_t)mls + 2), 0.0);
    double *Dl = lists.data(), *Al = lists.data() + mls + 2;
    for (int j = 0; j < D.nj; j++) for (int i = 0; i < D.ni; i++) {
      const size_t cc = (size_t)(i + D.ioff) + (size_t)(j + D.joff) * D.pitch;
      const int list_pos = (j + D.j_glob0) * D.ni_glob + (i + D.i_glob0) + 1;
      Dl[list_pos] = plane[cc] + p->Z_ref;
      Al[list_pos] = plane[D.slab + cc] * plane[2 * (size_t)D.slab + cc];
    }
    if (comm_nranks(c) > 1) {
      double *tmp = nullptr;
      HIPCHK(hipMalloc(&tmp, sizeof(double) * lists.size()));
      HIPCHK(hipMemcpyAsync(tmp, lists.data(), sizeof(double) * lists.size(), hipMemcpyHostToDevice, c->stream));
      const int rc = comm_allreduce_f64(c, tmp, lists.size(), 2);
      if (rc) { (void)hipFree(tmp); return rc; }
      HIPCHK(hipMemcpyAsync(lists.data(), tmp, sizeof(double) * lists.size(), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipFree(tmp));
    }
    std::vector<double> Dv(Dl, Dl + mls + 2), Av(Al, Al + mls + 2);
    create_depth_list(S, Dv, Av, mls);
  }
  S->lH.assign(nk, S->listsize - 1);                                             // :1194-1196
  return MOM6X_OK;
}

extern "C" int mom6x_depth_list(const mom6x_ctx *c, int *listsize, double *depth, double *area, double *vol_below) {
  REQUIRE(c && c->diag && listsize, MOM6X_EINVAL, "mom6x_depth_list: mom6x_sum_output_init has not been called");
  const DiagState *S = (const DiagState *)c->diag;
  *listsize = S->listsize;
  for (int n = 0; n < S->listsize; n++) {
    if (depth) depth[n] = S->depth[n + 1];
    if (area) area[n] = S->area[n + 1];
    if (vol_below) vol_below[n] = S->vol_below[n + 1];
  }
  return MOM6X_OK;
}

// The sums of write_energy :500-791; the bookkeeping around them (the schedule, the previous values, the files) is the host's
extern "C" int mom6x_write_energy(mom6x_ctx *c, const double *u, const double *v, const double *h, const double *T, const double *S_,
                                  mom6x_energy_sums *out, double *mass_lay, double *KE, double *PE, double *Z_0APE) {
  REQUIRE(c && u && v && h && out && mass_lay && KE && PE && Z_0APE, MOM6X_EINVAL, "mom6x_write_energy: null argument");
  REQUIRE(c->diag, MOM6X_EINVAL, "write_energy: Module must be initialized before it is used.");
  DiagState *S = (DiagState *)c->diag;
  REQUIRE(!S->p.use_temperature || (T && S_), MOM6X_EINVAL, "write_energy: ENABLE_THERMODYNAMICS needs tv%T and tv%S");
  const mom6x_dims &D = c->dims;
  const Dm d = c->d;
  const int nz = D.nk, ni = D.ni, nj = D.nj;
  const mom6x_vgrid &GV = c->GV;
  HostAcc H;
  int rc;
  memset(out, 0, sizeof(*out));
  // mass_tot = reproducing_sum(tmp1, sums=mass_lay, EFP_sum=mass_EFP) :509
  SrcMass sm; sm.h = h; sm.G = c->G; sm.H_to_RZ = GV.H_to_RZ;
  if ((rc = reduce_src(c, sm, "k_reduce_mass", nz, 0, ni - 1, 0, nj - 1, M_SUM, 0, 1.0, true, H))) return rc;
  if ((rc = reproducing_sum_finish(c, H, nz, 3, 1.0, 0, &out->mass_tot, mass_lay, out->mass_EFP, nullptr, nullptr))) return rc;
  out->PE_tot = 0.0;
  if (S->p.do_APE_calc) {
    std::vector<double> vol_lay(nz);
    for (int k = 0; k < nz; k++) vol_lay[k] = (1.0 / GV.Rho0) * mass_lay[k];      // :512
    int lbelow = 1, li; double volbelow = 0.0;                                    // :598-620
    for (int k = nz; k >= 1; k--) {
      volbelow = volbelow + vol_lay[k - 1];
      if ((volbelow >= S->vol_below[S->lH[k - 1]]) && (volbelow < S->vol_below[S->lH[k - 1] + 1])) li = S->lH[k - 1];
      else {
        int labove = S->listsize;
        li = (labove + lbelow) / 2;
        while (li > lbelow) {
          if (volbelow < S->vol_below[li]) labove = li; else lbelow = li;
          li = (labove + lbelow) / 2;
        }
        S->lH[k - 1] = li;
      }
      lbelow = li;
      Z_0APE[k - 1] = S->depth[li] - (volbelow - S->vol_below[li]) / S->area[li];
    }
    Z_0APE[nz] = S->depth[2];
    double *PE_pt;
    if ((rc = ctx_scratch(c, SCR_t0, nz + 1, &PE_pt))) return rc;
    HIPCHK(hipMemcpyAsync(S->z0_dev, Z_0APE, sizeof(double) * (nz + 1), hipMemcpyHostToDevice, c->stream));
    KLAUNCH(c, "k_pe_pt", k_pe_pt, dim3((ni + 255) / 256, nj), dim3(256), d, c->G, h, S->z0_dev, S->g_prime_dev, GV.Rho0, GV.H_to_Z,
            S->p.Z_ref, PE_pt);
    if ((rc = reduce_run(c, PE_pt, nz + 1, 0, ni - 1, 0, nj - 1, M_SUM, 0, 1.0, true, H))) return rc;
    if ((rc = reproducing_sum_finish(c, H, nz + 1, 3, 1.0, 0, &out->PE_tot, PE, nullptr, nullptr, nullptr))) return rc;
  } else {
    for (int K = 0; K <= nz; K++) { PE[K] = 0.0; Z_0APE[K] = 0.0; }
  }
  SrcKE sk; sk.h = h; sk.u = u; sk.v = v; sk.G = c->G; sk.H_to_RZ = GV.H_to_RZ;
  if ((rc = reduce_src(c, sk, "k_reduce_KE", nz, 0, ni - 1, 0, nj - 1, M_SUM, 0, 1.0, true, H))) return rc;
  if ((rc = reproducing_sum_finish(c, H, nz, 3, 1.0, 0, &out->KE_tot, KE, nullptr, nullptr, nullptr))) return rc;
  if (S->p.use_temperature) {
    double *ts;
    if ((rc = ctx_scratch(c, SCR_t1, 2, &ts))) return rc;
    KLAUNCH(c, "k_ts_int", k_ts_int, dim3((ni + 255) / 256, nj), dim3(256), d, c->G, h, T, S_, S->p.C_p, GV.H_to_RZ, ts, ts + D.slab);
    double dummy;
    if ((rc = reduce_run(c, ts, 1, 0, ni - 1, 0, nj - 1, M_SUM, 0, 1.0, true, H))) return rc;
    if ((rc = reproducing_sum_finish(c, H, 1, 2, 1.0, 0, &dummy, nullptr, out->salt_EFP, nullptr, nullptr))) return rc;
    if ((rc = reduce_run(c, ts + D.slab, 1, 0, ni - 1, 0, nj - 1, M_SUM, 0, 1.0, true, H))) return rc;
    if ((rc = reproducing_sum_finish(c, H, 1, 2, 1.0, 0, &dummy, nullptr, out->heat_EFP, nullptr, nullptr))) return rc;
  }
  HIPCHK(hipMemsetAsync(S->cfl_dev, 0, sizeof(long long) * 2, c->stream));
  KLAUNCH(c, "k_max_cfl", k_max_cfl, dim3((ni + 1 + 255) / 256, nj + 1, 1), dim3(256), d, c->G, u, v, S->p.dt_in_T, S->cfl_dev);
  if ((rc = comm_allreduce_i64(c, S->cfl_dev, 2, 1))) return rc;
  long long cb[2];
  HIPCHK(hipMemcpyAsync(cb, S->cfl_dev, sizeof(cb), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(&out->max_CFL[0], &cb[0], 8); memcpy(&out->max_CFL[1], &cb[1], 8);
  return MOM6X_OK;
}

void diag_sums_free(mom6x_ctx *c) {
  if (c->red) (void)hipFree(c->red);
  c->red = nullptr; c->red_cap = 0;
  diag_state_free((DiagState *)c->diag); c->diag = nullptr;
}
