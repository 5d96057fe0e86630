// halo.hip -- halo updates (pass_var / pass_vector / do_group_pass, MOM_domain_infra.F90:171-1190).
//
// Single-tile part: a re-entrant direction is a wrap copy from the tile's own opposite edge (what
// mpp_update_domains does on one PE); closed directions leave the halo untouched.  With MOM6's
// symmetric memory the u-point computational domain is I = -1..ni-1 (both ends owned by the tile), so
// only I <= -2 and I >= ni are halo points; likewise for v/q points in j.  A whole group pass (many
// fields) is one launch per direction.  x is wrapped before y so that corners are filled from wrapped
// data (To_All without Omit_Corners).
#include "mom6x_dev.h"

#define MAXF 16
struct WrapArgs { double *f[MAXF]; int stg[MAXF]; int nk[MAXF]; int n; };

__global__ void k_wrap_x(Dm d, WrapArgs A) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;   // row index within the data domain
  const int hh = threadIdx.y;                             // halo column 0..halo (extra one for B-points)
  const int k = blockIdx.z;
  const int w = d.halo, ni = d.ni;
  for (int m = 0; m < A.n; m++) {
    if (k >= A.nk[m]) continue;
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    const int j = -w - yB + jj;
    if (j > d.nj - 1 + w) continue;
    double *p = A.f[m] + (size_t)k * d.slab;
    if (hh < w) {   // west halo: i = -w-xB .. -1-xB   <- i+ni ; east halo: i = ni .. ni-1+w <- i-ni
      const int iw = -w - xB + hh;
      p[ix2(d, iw, j)] = p[ix2(d, iw + ni, j)];
      const int ie = ni + hh;
      p[ix2(d, ie, j)] = p[ix2(d, ie - ni, j)];
    }
  }
}

__global__ void k_wrap_y(Dm d, WrapArgs A) {
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  const int hh = threadIdx.y;
  const int k = blockIdx.z;
  const int w = d.halo, nj = d.nj;
  for (int m = 0; m < A.n; m++) {
    if (k >= A.nk[m]) continue;
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    const int i = -w - xB + ii;
    if (i > d.ni - 1 + w) continue;
    double *p = A.f[m] + (size_t)k * d.slab;
    if (hh < w) {
      const int js = -w - yB + hh;
      p[ix2(d, i, js)] = p[ix2(d, i, js + nj)];
      const int jn = nj + hh;
      p[ix2(d, i, jn)] = p[ix2(d, i, jn - nj)];
    }
  }
}

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n) {
  const Dm d = c->d;
  if (!c->dims.reentrant_x && !c->dims.reentrant_y) return;
  for (int base = 0; base < n; base += MAXF) {
    WrapArgs A;
    A.n = (n - base < MAXF) ? (n - base) : MAXF;
    int nkmax = 1;
    for (int m = 0; m < A.n; m++) {
      A.f[m] = fields[base + m]; A.stg[m] = staggers[base + m]; A.nk[m] = nks[base + m];
      if (A.nk[m] > nkmax) nkmax = A.nk[m];
    }
    const dim3 b(64, d.halo, 1);
    if (c->dims.reentrant_x) {
      const int rows = d.nj + 2 * d.halo + 1;
      KLAUNCH(c, "k_wrap_x", k_wrap_x, dim3((rows + 63) / 64, 1, nkmax), b, d, A);
    }
    if (c->dims.reentrant_y) {
      const int cols = d.ni + 2 * d.halo + 1;
      KLAUNCH(c, "k_wrap_y", k_wrap_y, dim3((cols + 63) / 64, 1, nkmax), b, d, A);
    }
  }
}
