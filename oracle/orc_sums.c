/* orc_sums.c -- ORACLE (test infrastructure, never shipped): the order-invariant sums and the checksums that the
 * reference's regression artefacts (ocean.stats, the debugging checksum lines, the restart `checksum` attributes)
 * are made of.  Restated from
 *   src/framework/MOM_coms.F90      reproducing_EFP_sum_2d :96, reproducing_sum_2d :235, reproducing_sum_3d :349,
 *                                   real_to_ints :561, ints_to_real :605, increment_ints :618, increment_ints_faster :652,
 *                                   carry_overflow :685, regularize_ints :709, EFP_plus :761, EFP_minus :772
 *   src/framework/MOM_checksums.F90 chksum_h_3d :1405 (and the 2-d / u / v / B variants), bitcount :2678,
 *                                   field_checksum_real_3d :2480
 * with the loop structure of the reference (one PE).  All integer work: the device must agree bit for bit.
 *
 * field_chksum is mpp_chksum of FMS, which is NOT in /root/reference (fetched by ac/deps/Makefile; pinned
 * FMS 2025.02.01 in .testing/Makefile:80): its published algorithm -- the wrapping 64-bit integer sum of the IEEE-754
 * bit patterns of the field, summed over PEs (mpp/include/mpp_chksum.h + mpp_chksum_int.h) -- is restated here.
 *
 * PARITY UNPINNED for this file: the reference holds no numbers for these routines (its regression tests compare
 * two runs of itself), so the restatement is checked by construction properties only (order invariance, exact
 * sums of exactly representable values, agreement of the EFP total with exact rational arithmetic in Python).
 */
#include <stdint.h>
#include "orc_common.h"

#define NI_EFP 6
static const int64_t prec = ((int64_t)1) << 46;                       /* :30 */
#define R_PREC 70368744177664.0                                        /* 2**46 :31 */
static const double I_prec = 1.0 / R_PREC;                             /* :32 */
static const int max_count_prec = (1 << (63 - 46)) - 1;                /* :33 */
static const double pr[NI_EFP] = {R_PREC * R_PREC, R_PREC, 1.0, 1.0 / R_PREC, (1.0 / R_PREC) / R_PREC,
                                  ((1.0 / R_PREC) / R_PREC) / R_PREC};                       /* :40-41 */
static const double I_pr[NI_EFP] = {(1.0 / R_PREC) / R_PREC, 1.0 / R_PREC, 1.0, R_PREC, R_PREC * R_PREC,
                                    R_PREC * R_PREC * R_PREC};                                /* :43-44 */
#define MAX_EFP_FLOAT (R_PREC * R_PREC * 9223372036854775808.0)       /* pr(1) * (2.**63 - 1.) :46 (rounds to 2**63) */

static int overflow_error = 0, NaN_error = 0;                          /* module variables :51-52 */

static int is_nan_like(double r) { return ((r >= 1e30) == (r < 1e30)); }   /* the reference's NaN test :665 */

/* real_to_ints :561-601 (prec_error < 0: absent).  Returns 1 where the reference calls MOM_error(FATAL). */
int orc_real_to_ints(double r, int64_t prec_error, int *overflow, int64_t *ints) {
  const int64_t prec_err = (prec_error >= 0) ? prec_error : prec;
  for (int i = 0; i < NI_EFP; i++) ints[i] = 0;
  if (is_nan_like(r)) { NaN_error = 1; return 0; }
  const int sgn = (r < 0.0) ? -1 : 1;
  double rs = fabs(r);
  if (overflow) {
    if (!(rs < (double)prec_err * pr[0])) *overflow = 1;
  } else if (!(rs < (double)prec_err * pr[0])) return 1;
  for (int i = 0; i < NI_EFP; i++) {
    const int64_t ival = (int64_t)(rs * I_pr[i]);
    rs = rs - (double)ival * pr[i];
    ints[i] = sgn * ival;
  }
  return 0;
}

double orc_ints_to_real(const int64_t *ints) {                         /* :605-614 */
  double r = 0.0;
  for (int i = 0; i < NI_EFP; i++) r = r + pr[i] * (double)ints[i];
  return r;
}

/* increment_ints :618-648 (prec_error < 0: absent) */
void orc_increment_ints(int64_t *int_sum, const int64_t *int2, int64_t prec_error) {
  for (int i = NI_EFP - 1; i >= 1; i--) {
    int_sum[i] = int_sum[i] + int2[i];
    if (int_sum[i] > prec) { int_sum[i] = int_sum[i] - prec; int_sum[i - 1] = int_sum[i - 1] + 1; }
    else if (int_sum[i] < -prec) { int_sum[i] = int_sum[i] + prec; int_sum[i - 1] = int_sum[i - 1] - 1; }
  }
  int_sum[0] = int_sum[0] + int2[0];
  const int64_t lim = (prec_error >= 0) ? prec_error : prec;
  if (llabs(int_sum[0]) > lim) overflow_error = 1;
}

static void increment_ints_faster(int64_t *int_sum, double r, double *max_mag_term) {   /* :652-682 */
  if (is_nan_like(r)) { NaN_error = 1; return; }
  const int sgn = (r < 0.0) ? -1 : 1;
  double rs = fabs(r);
  if (rs > fabs(*max_mag_term)) *max_mag_term = r;
  if (rs > MAX_EFP_FLOAT) { overflow_error = 1; return; }
  for (int i = 0; i < NI_EFP; i++) {
    const int64_t ival = (int64_t)(rs * I_pr[i]);
    rs = rs - (double)ival * pr[i];
    int_sum[i] = int_sum[i] + sgn * ival;
  }
}

static void carry_overflow(int64_t *int_sum, int64_t prec_error) {     /* :685-705 */
  for (int i = NI_EFP - 1; i >= 1; i--) if (llabs(int_sum[i]) >= prec) {
    const int num_carry = (int)((double)int_sum[i] * I_prec);          /* a default integer; the product is a real */
    int_sum[i] = int_sum[i] - (int64_t)num_carry * prec;
    int_sum[i - 1] = int_sum[i - 1] + num_carry;
  }
  if (llabs(int_sum[0]) > prec_error) overflow_error = 1;
}

void orc_regularize_ints(int64_t *int_sum) {                           /* :709-747 */
  for (int i = NI_EFP - 1; i >= 1; i--) if (llabs(int_sum[i]) >= prec) {
    const int num_carry = (int)((double)int_sum[i] * I_prec);
    int_sum[i] = int_sum[i] - (int64_t)num_carry * prec;
    int_sum[i - 1] = int_sum[i - 1] + num_carry;
  }
  int positive = 1;
  for (int i = 0; i < NI_EFP; i++) if (llabs(int_sum[i]) > 0) { if (int_sum[i] < 0) positive = 0; break; }
  if (positive) {
    for (int i = NI_EFP - 1; i >= 1; i--) if (int_sum[i] < 0) { int_sum[i] = int_sum[i] + prec; int_sum[i - 1] = int_sum[i - 1] - 1; }
  } else {
    for (int i = NI_EFP - 1; i >= 1; i--) if (int_sum[i] > 0) { int_sum[i] = int_sum[i] - prec; int_sum[i - 1] = int_sum[i - 1] + 1; }
  }
}

void orc_EFP_plus(const int64_t *a, const int64_t *b, int64_t *out) {  /* :761 */
  for (int i = 0; i < NI_EFP; i++) out[i] = a[i];
  orc_increment_ints(out, b, -1);
}
void orc_EFP_minus(const int64_t *a, const int64_t *b, int64_t *out) { /* :772 */
  for (int i = 0; i < NI_EFP; i++) out[i] = -1 * b[i];
  orc_increment_ints(out, a, -1);
}
double orc_EFP_to_real(int64_t *a) { orc_regularize_ints(a); return orc_ints_to_real(a); }   /* :797 */

static const int64_t PREC_ERROR_1PE = (((int64_t)1) << 62) + ((((int64_t)1) << 62) - 1);    /* :137 with one PE */

/* reproducing_EFP_sum_2d :96-228 over the points (is..ie, js..je) of one pitched 2-D plane (local compute indices).
 * err == NULL: returns 1 (NaN), 2 (conversion overflow) or 3 (overflow) where the reference stops with FATAL. */
int orc_reproducing_EFP_sum_2d(const mom6x_dims *d, const double *array, int is, int ie, int js, int je, int overflow_check,
                               double unscale, int64_t *EFP_sum, int *err) {
  const int64_t prec_error = PREC_ERROR_1PE;
  const int do_unscale = (unscale != 1.0);
  const double descale = do_unscale ? unscale : 1.0;
  int64_t ints_sum[NI_EFP] = {0, 0, 0, 0, 0, 0};
  double max_mag_term = 0.0;
  overflow_error = 0; NaN_error = 0;
  if (overflow_check) {
    if ((je + 1 - js) * (ie + 1 - is) < max_count_prec) {
      for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
        increment_ints_faster(ints_sum, do_unscale ? unscale * array[IX2(d, i, j)] : array[IX2(d, i, j)], &max_mag_term);
      carry_overflow(ints_sum, prec_error);
    } else if ((ie + 1 - is) < max_count_prec) {
      for (int j = js; j <= je; j++) {
        for (int i = is; i <= ie; i++) increment_ints_faster(ints_sum, descale * array[IX2(d, i, j)], &max_mag_term);
        carry_overflow(ints_sum, prec_error);
      }
    } else {
      for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        int64_t t[NI_EFP];
        if (orc_real_to_ints(descale * array[IX2(d, i, j)], prec_error, NULL, t)) return 2;
        orc_increment_ints(ints_sum, t, prec_error);
      }
    }
  } else {
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      const int sgn = (array[IX2(d, i, j)] < 0.0) ? -1 : 1;
      double rs = fabs(descale * array[IX2(d, i, j)]);
      for (int n = 0; n < NI_EFP; n++) {
        const int64_t ival = (int64_t)(rs * I_pr[n]);
        rs = rs - (double)ival * pr[n];
        ints_sum[n] = ints_sum[n] + sgn * ival;
      }
    }
    carry_overflow(ints_sum, prec_error);
  }
  if (err) {
    *err = 0;
    if (overflow_error) *err += 2;
    if (NaN_error) *err += 4;
    if (*err > 0) for (int n = 0; n < NI_EFP; n++) ints_sum[n] = 0;
  } else {
    if (NaN_error) return 1;
    if (fabs(max_mag_term) >= (double)prec_error * pr[0]) return 2;
    if (overflow_error) return 3;
  }
  orc_regularize_ints(ints_sum);
  for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = ints_sum[n];
  return 0;
}

/* reproducing_sum_2d :235-343 with reproducing = .true. */
int orc_reproducing_sum_2d(const mom6x_dims *d, const double *array, int is, int ie, int js, int je, double unscale,
                           double *sum, int64_t *EFP_sum, int *err) {
  int64_t v[NI_EFP];
  double I_unscale = 1.0;
  if (unscale != 1.0 && fabs(unscale) > 0.0) I_unscale = 1.0 / unscale;
  const int rc = orc_reproducing_EFP_sum_2d(d, array, is, ie, js, je, 1, unscale, v, err);
  if (rc) return rc;
  *sum = orc_ints_to_real(v) * I_unscale;
  if (EFP_sum) for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = v[n];
  return 0;
}

/* reproducing_sum_3d :349-558; array: nk pitched planes. sums / EFP_sum / EFP_lay_sums / err may be NULL. */
int orc_reproducing_sum_3d(const mom6x_dims *d, const double *array, int nk, int is, int ie, int js, int je, double unscale,
                           double *sum, double *sums, int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err) {
  const int64_t prec_error = PREC_ERROR_1PE;
  const int jsz = je + 1 - js, isz = ie + 1 - is;
  const int do_unscale = (unscale != 1.0);
  const double descale = do_unscale ? unscale : 1.0;
  double max_mag_term = 0.0;
  overflow_error = 0; NaN_error = 0;
  if (sums || EFP_lay_sums) {
    int64_t *ints_sums = (int64_t *)calloc((size_t)NI_EFP * (size_t)nk, sizeof(int64_t));
    if (jsz * isz < max_count_prec) {
      for (int k = 0; k < nk; k++) {
        for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
          increment_ints_faster(ints_sums + NI_EFP * k, do_unscale ? unscale * array[IX3(d, i, j, k)] : array[IX3(d, i, j, k)], &max_mag_term);
        carry_overflow(ints_sums + NI_EFP * k, prec_error);
      }
    } else if (isz < max_count_prec) {
      for (int k = 0; k < nk; k++) for (int j = js; j <= je; j++) {
        for (int i = is; i <= ie; i++) increment_ints_faster(ints_sums + NI_EFP * k, descale * array[IX3(d, i, j, k)], &max_mag_term);
        carry_overflow(ints_sums + NI_EFP * k, prec_error);
      }
    } else {
      for (int k = 0; k < nk; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        int64_t t[NI_EFP];
        if (orc_real_to_ints(descale * array[IX3(d, i, j, k)], prec_error, NULL, t)) { free(ints_sums); return 2; }
        orc_increment_ints(ints_sums + NI_EFP * k, t, prec_error);
      }
    }
    if (err) {
      *err = 0;
      if (fabs(max_mag_term) >= (double)prec_error * pr[0]) *err += 1;
      if (overflow_error) *err += 2;
      if (NaN_error) *err += 2;
      if (*err > 0) for (int n = 0; n < NI_EFP * nk; n++) ints_sums[n] = 0;
    } else {
      int rc = 0;
      if (NaN_error) rc = 1;
      else if (fabs(max_mag_term) >= (double)prec_error * pr[0]) rc = 2;
      else if (overflow_error) rc = 3;
      if (rc) { free(ints_sums); return rc; }
    }
    double s = 0.0;
    for (int k = 0; k < nk; k++) {
      orc_regularize_ints(ints_sums + NI_EFP * k);
      const double val = orc_ints_to_real(ints_sums + NI_EFP * k);
      if (sums) sums[k] = val;
      s = s + val;
    }
    if (EFP_lay_sums) for (int n = 0; n < NI_EFP * nk; n++) EFP_lay_sums[n] = ints_sums[n];
    if (EFP_sum) {
      for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = 0;
      for (int k = 0; k < nk; k++) orc_increment_ints(EFP_sum, ints_sums + NI_EFP * k, -1);
    }
    *sum = s;
    free(ints_sums);
  } else {
    int64_t ints_sum[NI_EFP] = {0, 0, 0, 0, 0, 0};
    if (jsz * isz < max_count_prec) {
      for (int k = 0; k < nk; k++) {
        for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
          increment_ints_faster(ints_sum, do_unscale ? unscale * array[IX3(d, i, j, k)] : array[IX3(d, i, j, k)], &max_mag_term);
        carry_overflow(ints_sum, prec_error);
      }
    } else if (isz < max_count_prec) {
      for (int k = 0; k < nk; k++) for (int j = js; j <= je; j++) {
        for (int i = is; i <= ie; i++) increment_ints_faster(ints_sum, descale * array[IX3(d, i, j, k)], &max_mag_term);
        carry_overflow(ints_sum, prec_error);
      }
    } else {
      for (int k = 0; k < nk; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        int64_t t[NI_EFP];
        if (orc_real_to_ints(descale * array[IX3(d, i, j, k)], prec_error, NULL, t)) return 2;
        orc_increment_ints(ints_sum, t, prec_error);
      }
    }
    if (err) {
      *err = 0;
      if (fabs(max_mag_term) >= (double)prec_error * pr[0]) *err += 1;
      if (overflow_error) *err += 2;
      if (NaN_error) *err += 2;
      if (*err > 0) for (int n = 0; n < NI_EFP; n++) ints_sum[n] = 0;
    } else {
      if (NaN_error) return 1;
      if (fabs(max_mag_term) >= (double)prec_error * pr[0]) return 2;
      if (overflow_error) return 3;
    }
    orc_regularize_ints(ints_sum);
    *sum = orc_ints_to_real(ints_sum);
    if (EFP_sum) for (int n = 0; n < NI_EFP; n++) EFP_sum[n] = ints_sum[n];
  }
  if (do_unscale) {
    double I_unscale = 0.0;
    if (fabs(unscale) > 0.0) I_unscale = 1.0 / unscale;
    *sum = *sum * I_unscale;
    if (sums) for (int k = 0; k < nk; k++) sums[k] = sums[k] * I_unscale;
  }
  return 0;
}

/* ---- MOM_checksums.F90 ------------------------------------------------------------------------------------------- */
static const int bc_modulus = 1000000000;                              /* :110 */

static int bitcount(double x) {                                        /* :2678-2685 */
  uint64_t b; memcpy(&b, &x, 8);
  return __builtin_popcountll(b);
}

/* subchk of chksum_*_3d (:1541, :1733, :1937, ...): the h-point computational domain shifted by (di, dj), whatever the
 * staggering of the array.  `subchk` is a default integer: the running sum wraps like 32-bit two's complement. */
int orc_subchk(const mom6x_dims *d, const double *array, int nk, int di, int dj, double unscale) {
  uint32_t s = 0;
  for (int k = 0; k < nk; k++) for (int j = dj; j <= d->nj - 1 + dj; j++) for (int i = di; i <= d->ni - 1 + di; i++)
    s += (uint32_t)bitcount(fabs(unscale * array[IX3(d, i, j, k)]));
  return (int)((int32_t)s % bc_modulus);                               /* Fortran mod: the sign of the dividend, like C */
}

/* chksum_{h,u,v,B}_{2d,3d}: stagger 0 h :1413 (2-d :387), 1 u :1782 (2-d :1005), 2 v :1986 (2-d :1209), 3 B :1586
 * (2-d :688, whose shifts differ from the 3-d routine's: rank = 2 or 3 selects).  out: mean, min, max (as printed by chk_sum_msg3 :2638: 0. + value), bc0 and up to four shifted bitcounts
 * in the order the message prints them (sw se nw ne | N S E W | W | S); returns the number of shifted counts, -1 on NaN. */
int orc_chksum(const mom6x_dims *d, const double *array, int nk, int rank, int stagger, int haloshift, int symmetric, int omit_corners,
               int scale_present, double scale, double *stats, int *bc) {
  const int ni = d->ni, nj = d->nj;
  const double scaling = scale;
  for (int k = 0; k < nk; k++) for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++)
    if (array[IX3(d, i, j, k)] != array[IX3(d, i, j, k)]) return -1;   /* checkForNaNs :1453: the h-point domain */
  int sym_stats = symmetric;
  if (stagger != 0 && haloshift > 0) sym_stats = 1;
  /* subStats: min / max over the staggered domain, mean = reproducing_sum over the h-point domain / n */
  int Is = 0, Js = 0;
  if ((stagger == 1 || stagger == 3) && sym_stats) Is = -1;
  if ((stagger == 2 || stagger == 3) && sym_stats) Js = -1;
  const double f = scale_present ? scaling : 1.0;
  const int mul = scale_present;
  double aMin = mul ? f * array[IX3(d, 0, 0, 0)] : array[IX3(d, 0, 0, 0)], aMax = aMin;
  for (int k = 0; k < nk; k++) for (int j = Js; j < nj; j++) for (int i = Is; i < ni; i++) {
    const double v = mul ? f * array[IX3(d, i, j, k)] : array[IX3(d, i, j, k)];
    aMin = orc_min(aMin, v); aMax = orc_max(aMax, v);
  }
  /* the mean sums the (rescaled) array itself: reproducing_sum without unscale */
  double sum;
  double *tmp = NULL;
  const double *src = array;
  if (mul) {
    tmp = (double *)calloc((size_t)d->slab * (size_t)nk, sizeof(double));
    for (int k = 0; k < nk; k++) for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) tmp[IX3(d, i, j, k)] = f * array[IX3(d, i, j, k)];
    src = tmp;
  }
  int rc = orc_reproducing_sum_3d(d, src, nk, 0, ni - 1, 0, nj - 1, 1.0, &sum, NULL, NULL, NULL, NULL);
  if (tmp) free(tmp);
  if (rc) return -1;
  const int n = ni * nj * nk;
  stats[0] = sum / (double)n; stats[1] = 0. + aMin; stats[2] = 0. + aMax;
  const int hs = haloshift;
  bc[0] = orc_subchk(d, array, nk, 0, 0, scaling);
  const int sx = (stagger == 1 || stagger == 3), sy = (stagger == 2 || stagger == 3);   /* staggered in x / y */
  if (stagger == 0) {
    if (hs == 0) return 0;
    if (!omit_corners) {
      bc[1] = orc_subchk(d, array, nk, -hs, -hs, scaling); bc[2] = orc_subchk(d, array, nk, hs, -hs, scaling);
      bc[3] = orc_subchk(d, array, nk, -hs, hs, scaling); bc[4] = orc_subchk(d, array, nk, hs, hs, scaling);
    } else {
      bc[1] = orc_subchk(d, array, nk, 0, hs, scaling); bc[2] = orc_subchk(d, array, nk, 0, -hs, scaling);     /* N S E W */
      bc[3] = orc_subchk(d, array, nk, hs, 0, scaling); bc[4] = orc_subchk(d, array, nk, -hs, 0, scaling);
    }
    return 4;
  }
  if (hs == 0 && !symmetric) return 0;
  if (stagger == 1 || stagger == 2) {
    /* u :1870-1900 and v :2075-2105: the symmetric point lies one to the west (u) / south (v) */
    const int wx = (stagger == 1 && symmetric) ? 1 : 0, wy = (stagger == 2 && symmetric) ? 1 : 0;
    if (hs == 0) {
      bc[1] = orc_subchk(d, array, nk, sx ? -hs - 1 : 0, sy ? -hs - 1 : 0, scaling);   /* "W=" or "S=" */
      return 1;
    }
    if (!omit_corners) {
      bc[1] = orc_subchk(d, array, nk, -hs - wx, -hs - wy, scaling);                   /* sw */
      bc[2] = orc_subchk(d, array, nk, hs, -hs - wy, scaling);                         /* se */
      bc[3] = orc_subchk(d, array, nk, -hs - wx, hs, scaling);                         /* nw */
      bc[4] = orc_subchk(d, array, nk, hs, hs, scaling);                               /* ne */
    } else {
      bc[1] = orc_subchk(d, array, nk, 0, hs, scaling);                                /* N */
      bc[2] = orc_subchk(d, array, nk, 0, -hs - wy, scaling);                          /* S */
      bc[3] = orc_subchk(d, array, nk, hs, 0, scaling);                                /* E */
      bc[4] = orc_subchk(d, array, nk, -hs - wx, 0, scaling);                          /* W */
    }
    return 4;
  }
  if (rank == 2) {   /* chksum_B_2d :767-788 */
    if (!omit_corners) {
      const int w = symmetric ? 1 : 0;
      bc[1] = orc_subchk(d, array, nk, -hs - w, -hs - w, scaling); bc[2] = orc_subchk(d, array, nk, hs, -hs - w, scaling);
      bc[3] = orc_subchk(d, array, nk, -hs - w, hs, scaling); bc[4] = orc_subchk(d, array, nk, hs, hs, scaling);
    } else {
      bc[1] = orc_subchk(d, array, nk, 0, hs, scaling); bc[2] = orc_subchk(d, array, nk, 0, -hs, scaling);
      bc[3] = orc_subchk(d, array, nk, hs, 0, scaling); bc[4] = orc_subchk(d, array, nk, -hs, 0, scaling);
    }
    return 4;
  }
  /* chksum_B_3d :1665-1690: both branches of `sym` shift by -hshift-1 at the corners */
  if (!omit_corners) {
    bc[1] = orc_subchk(d, array, nk, -hs - 1, -hs - 1, scaling); bc[2] = orc_subchk(d, array, nk, hs, -hs - 1, scaling);
    bc[3] = orc_subchk(d, array, nk, -hs - 1, hs, scaling); bc[4] = orc_subchk(d, array, nk, hs, hs, scaling);
  } else {
    const int w = symmetric ? 1 : 0;
    bc[1] = orc_subchk(d, array, nk, 0, hs, scaling); bc[2] = orc_subchk(d, array, nk, 0, -hs - w, scaling);
    bc[3] = orc_subchk(d, array, nk, hs, 0, scaling); bc[4] = orc_subchk(d, array, nk, -hs - w, 0, scaling);
  }
  return 4;
}

/* field_checksum_real_3d :2480 -> field_chksum -> mpp_chksum (FMS, see the header): the wrapping int64 sum of the bit
 * patterns of unscale*field over (is..ie, js..je, all k). */
int64_t orc_field_chksum(const mom6x_dims *d, const double *array, int nk, int is, int ie, int js, int je, double unscale) {
  uint64_t s = 0;
  const int do_unscale = (unscale != 1.0);
  for (int k = 0; k < nk; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    const double v = do_unscale ? unscale * array[IX3(d, i, j, k)] : array[IX3(d, i, j, k)];
    uint64_t b; memcpy(&b, &v, 8);
    s += b;
  }
  return (int64_t)s;
}

/* ---- MOM_sum_output.F90 ------------------------------------------------------------------------------------------ */
/* create_depth_list :1203-1326 (one tile = the global domain).  depth/area/vol_below: 1-based arrays with room for
 * ni*nj + 2 entries; returns DL%listsize. */
int orc_create_depth_list(const mom6x_dims *d, const double *G, double Z_ref, double min_depth_inc,
                          double *depth, double *area_out, double *vol_below) {
  const int niglobal = d->ni_glob, mls = d->ni_glob * d->nj_glob;
  double *Dlist = (double *)calloc((size_t)mls + 2, sizeof(double)), *AreaList = (double *)calloc((size_t)mls + 2, sizeof(double));
  int *indx2 = (int *)calloc((size_t)mls + 2, sizeof(int));
  const double *bathyT = GM(G, d, MOM6X_G_bathyT), *mask = GM(G, d, MOM6X_G_mask2dT), *areaT = GM(G, d, MOM6X_G_areaT);
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    const int list_pos = (j + d->j_glob0) * niglobal + (i + d->i_glob0) + 1;
    Dlist[list_pos] = bathyT[IX2(d, i, j)] + Z_ref;
    AreaList[list_pos] = mask[IX2(d, i, j)] * areaT[IX2(d, i, j)];
  }
  for (int j = 1; j <= mls + 1; j++) indx2[j] = j;
  int k = mls / 2 + 1, ir = mls, indxt, i, j;
  double Dnow;
  for (;;) {                                                           /* the heap sort of :1246-1267 */
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
  for (k = mls - 1; k >= 1; k--) if (Dlist[indx2[k]] < D_list_prev - min_depth_inc) { list_size = list_size + 1; D_list_prev = Dlist[indx2[k]]; }
  const int listsize = list_size + 1;
  double vol = 0.0, area = 0.0, Dprev = Dlist[indx2[mls]];
  D_list_prev = Dprev;
  int kl = 0;
  for (k = mls; k >= 1; k--) {
    i = indx2[k];
    vol = vol + area * (Dprev - Dlist[i]);
    area = area + AreaList[i];
    int add_to_list = 0;
    if (kl == 0 || k == 1) add_to_list = 1;
    else if (Dlist[indx2[k - 1]] < D_list_prev - min_depth_inc) { add_to_list = 1; D_list_prev = Dlist[indx2[k - 1]]; }
    if (add_to_list) { kl = kl + 1; depth[kl] = Dlist[i]; area_out[kl] = area; vol_below[kl] = vol; }
    Dprev = Dlist[i];
  }
  while (kl + 1 < listsize) {
    kl = kl + 1;
    vol_below[kl] = vol_below[kl - 1] * 1.000001; area_out[kl] = area_out[kl - 1]; depth[kl] = depth[kl - 1];
  }
  vol_below[listsize] = vol_below[listsize - 1] * 1000.0;
  area_out[listsize] = area_out[listsize - 1];
  depth[listsize] = depth[listsize - 1];
  free(Dlist); free(AreaList); free(indx2);
  return listsize;
}

/* The sums of write_energy :598-791 (Boussinesq, unit scaling factors of 1): layer masses, the zero-APE depths of the
 * interfaces from the depth list, the interface APE, the layer KE, the salt and heat content, the two CFL maxima.
 * lH: the search hints CS%lH (1-based values, nk of them), updated as in the reference.  T, S may be NULL.
 * scal: mass_tot, KE_tot, PE_tot, max_CFL(1), max_CFL(2).  Returns nonzero where the reference stops. */
int orc_write_energy(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *g_prime, int do_APE_calc,
                     double dt_in_T, double Z_ref, double C_p, int listsize, const double *DL_depth, const double *DL_area,
                     const double *DL_vol_below, int *lH, const double *u, const double *v, const double *h, const double *T,
                     const double *S, double *mass_lay, double *KE, double *PE, double *Z_0APE, double *scal, int64_t *mass_EFP,
                     int64_t *salt_EFP, int64_t *heat_EFP) {
  const int nz = d->nk, ni = d->ni, nj = d->nj;
  const double *mask = GM(G, d, MOM6X_G_mask2dT), *areaT = GM(G, d, MOM6X_G_areaT), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double *IareaT = GM(G, d, MOM6X_G_IareaT), *dy_Cu = GM(G, d, MOM6X_G_dy_Cu), *dx_Cv = GM(G, d, MOM6X_G_dx_Cv);
  const double *IdxCu = GM(G, d, MOM6X_G_IdxCu), *IdyCv = GM(G, d, MOM6X_G_IdyCv);
  double *areaTm = (double *)calloc((size_t)d->slab, sizeof(double));
  double *tmp1 = (double *)calloc((size_t)d->slab * (size_t)(nz + 1), sizeof(double));
  int rc;
  for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) areaTm[IX2(d, i, j)] = mask[IX2(d, i, j)] * areaT[IX2(d, i, j)];
  for (int k = 0; k < nz; k++) for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++)
    tmp1[IX3(d, i, j, k)] = h[IX3(d, i, j, k)] * (GV->H_to_RZ * areaTm[IX2(d, i, j)]);
  double mass_tot;
  if ((rc = orc_reproducing_sum_3d(d, tmp1, nz, 0, ni - 1, 0, nj - 1, 1.0, &mass_tot, mass_lay, mass_EFP, NULL, NULL))) goto done;
  double PE_tot = 0.0;
  if (do_APE_calc) {
    double *vol_lay = (double *)calloc((size_t)nz, sizeof(double));
    for (int k = 0; k < nz; k++) vol_lay[k] = (1.0 / GV->Rho0) * mass_lay[k];
    int lbelow = 1, li; double volbelow = 0.0;                         /* :598-620, 1-based list indices */
    for (int k = nz; k >= 1; k--) {
      volbelow = volbelow + vol_lay[k - 1];
      if ((volbelow >= DL_vol_below[lH[k - 1]]) && (volbelow < DL_vol_below[lH[k - 1] + 1])) li = lH[k - 1];
      else {
        int labove = listsize;
        li = (labove + lbelow) / 2;
        while (li > lbelow) {
          if (volbelow < DL_vol_below[li]) labove = li; else lbelow = li;
          li = (labove + lbelow) / 2;
        }
        lH[k - 1] = li;
      }
      lbelow = li;
      Z_0APE[k - 1] = DL_depth[li] - (volbelow - DL_vol_below[li]) / DL_area[li];
    }
    Z_0APE[nz] = DL_depth[2];
    free(vol_lay);
    memset(tmp1, 0, sizeof(double) * (size_t)d->slab * (size_t)(nz + 1));
    for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) {         /* :623-634 */
      double hbelow = 0.0;
      for (int K = nz; K >= 1; K--) {
        hbelow = hbelow + h[IX3(d, i, j, K - 1)] * GV->H_to_Z;
        const double hint = Z_0APE[K - 1] + (hbelow - (bathyT[IX2(d, i, j)] + Z_ref));
        double hbot = Z_0APE[K - 1] - (bathyT[IX2(d, i, j)] + Z_ref);
        hbot = (hbot + fabs(hbot)) * 0.5;
        tmp1[IX3(d, i, j, K - 1)] = (0.5 * areaTm[IX2(d, i, j)]) * (GV->Rho0 * g_prime[K - 1]) * (hint * hint - hbot * hbot);
      }
    }
    if ((rc = orc_reproducing_sum_3d(d, tmp1, nz + 1, 0, ni - 1, 0, nj - 1, 1.0, &PE_tot, PE, NULL, NULL, NULL))) goto done;
  } else {
    for (int K = 0; K <= nz; K++) { PE[K] = 0.0; Z_0APE[K] = 0.0; }
  }
  memset(tmp1, 0, sizeof(double) * (size_t)d->slab * (size_t)(nz + 1));
  for (int k = 0; k < nz; k++) for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) {   /* :669-673 */
    const double uw = u[IX3(d, i - 1, j, k)], ue = u[IX3(d, i, j, k)], vs = v[IX3(d, i, j - 1, k)], vn = v[IX3(d, i, j, k)];
    tmp1[IX3(d, i, j, k)] = (0.25 * GV->H_to_RZ * (areaTm[IX2(d, i, j)] * h[IX3(d, i, j, k)])) *
                            (((uw * uw) + (ue * ue)) + ((vs * vs) + (vn * vn)));
  }
  double KE_tot;
  if ((rc = orc_reproducing_sum_3d(d, tmp1, nz, 0, ni - 1, 0, nj - 1, 1.0, &KE_tot, KE, NULL, NULL, NULL))) goto done;
  if (T && S) {                                                        /* :677-686 */
    double *Temp_int = (double *)calloc((size_t)d->slab, sizeof(double)), *Salt_int = (double *)calloc((size_t)d->slab, sizeof(double));
    for (int k = 0; k < nz; k++) for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) {
      const double hm = h[IX3(d, i, j, k)] * (GV->H_to_RZ * areaTm[IX2(d, i, j)]);
      Salt_int[IX2(d, i, j)] = Salt_int[IX2(d, i, j)] + S[IX3(d, i, j, k)] * hm;
      Temp_int[IX2(d, i, j)] = Temp_int[IX2(d, i, j)] + (C_p * T[IX3(d, i, j, k)]) * hm;
    }
    rc = orc_reproducing_EFP_sum_2d(d, Salt_int, 0, ni - 1, 0, nj - 1, 1, 1.0, salt_EFP, NULL);
    if (!rc) rc = orc_reproducing_EFP_sum_2d(d, Temp_int, 0, ni - 1, 0, nj - 1, 1, 1.0, heat_EFP, NULL);
    free(Temp_int); free(Salt_int);
    if (rc) goto done;
  }
  double max_CFL[2] = {0.0, 0.0};                                      /* :701-722 */
  for (int k = 0; k < nz; k++) for (int j = 0; j < nj; j++) for (int I = -1; I < ni; I++) {
    double CFL_Iarea = IareaT[IX2(d, I, j)];
    if (u[IX3(d, I, j, k)] < 0.0) CFL_Iarea = IareaT[IX2(d, I + 1, j)];
    const double CFL_trans = fabs(u[IX3(d, I, j, k)] * dt_in_T) * (dy_Cu[IX2(d, I, j)] * CFL_Iarea);
    const double CFL_lin = fabs(u[IX3(d, I, j, k)] * dt_in_T) * IdxCu[IX2(d, I, j)];
    max_CFL[0] = orc_max(max_CFL[0], CFL_trans); max_CFL[1] = orc_max(max_CFL[1], CFL_lin);
  }
  for (int k = 0; k < nz; k++) for (int J = -1; J < nj; J++) for (int i = 0; i < ni; i++) {
    double CFL_Iarea = IareaT[IX2(d, i, J)];
    if (v[IX3(d, i, J, k)] < 0.0) CFL_Iarea = IareaT[IX2(d, i, J + 1)];
    const double CFL_trans = fabs(v[IX3(d, i, J, k)] * dt_in_T) * (dx_Cv[IX2(d, i, J)] * CFL_Iarea);
    const double CFL_lin = fabs(v[IX3(d, i, J, k)] * dt_in_T) * IdyCv[IX2(d, i, J)];
    max_CFL[0] = orc_max(max_CFL[0], CFL_trans); max_CFL[1] = orc_max(max_CFL[1], CFL_lin);
  }
  scal[0] = mass_tot; scal[1] = KE_tot; scal[2] = PE_tot; scal[3] = max_CFL[0]; scal[4] = max_CFL[1];
  rc = 0;
done:
  free(areaTm); free(tmp1);
  return rc;
}
