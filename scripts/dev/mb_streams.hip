// Dev microbenchmark: one kernel that reads NIN arrays and writes NOUT arrays at the same index (the shape of k_corad_lds / k_pgf_main:
// many concurrent streams over arrays of one size), with the arrays' start addresses as hipMalloc hands them out and skewed against each
// other by m * SKEW bytes.  Question: does the relative placement of equally sized arrays cost bandwidth on this box?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_streams scripts/dev/mb_streams.hip && /tmp/mb_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int NIN, int NOUT> struct P { const double *in[NIN]; double *out[NOUT]; };
template <int NIN, int NOUT>
__global__ void __launch_bounds__(256) k(P<NIN, NOUT> p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.;
#pragma unroll
  for (int m = 0; m < NIN; m++) s += p.in[m][i];
#pragma unroll
  for (int m = 0; m < NOUT; m++) p.out[m][i] = s * (m + 1);
}
template <int NIN, int NOUT>
int run(std::vector<char *> &base, size_t n, size_t skew, const char *tag) {
  P<NIN, NOUT> p;
  for (int m = 0; m < NIN; m++) p.in[m] = (const double *)(base[m] + (size_t)m * skew);
  for (int m = 0; m < NOUT; m++) p.out[m] = (double *)(base[NIN + m] + (size_t)(NIN + m) * skew);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 6; r++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<NIN, NOUT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
  }
  printf("%-8s in %2d out %2d skew %9zu B : %7.3f ms  %7.1f GB/s\n", tag, NIN, NOUT, skew, best, (NIN + NOUT) * 8.0 * n / best / 1e6);
  return 0;
}
int main() {
  const size_t n = (size_t)1448 * 1088 * 75, pad = (size_t)64 << 20;
  std::vector<char *> base(16);
  for (auto &b : base) { CK(hipMalloc(&b, n * 8 + pad)); CK(hipMemset(b, 0, n * 8 + pad)); }
  for (size_t m = 0; m < base.size(); m++) printf("array %2zu at %p  (offset from array 0: %lld MiB + %lld B)\n", m, (void *)base[m],
      (long long)((base[m] - base[0]) >> 20), (long long)((base[m] - base[0]) & ((1 << 20) - 1)));
  for (size_t skew : { (size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)((1 << 20) + 4096 + 256), (size_t)0 }) {
    if (run<2, 1>(base, n, skew, "triad")) return 1;
    if (run<9, 6>(base, n, skew, "corad")) return 1;
    if (run<7, 5>(base, n, skew, "pgf")) return 1;
    if (run<8, 0 + 1>(base, n, skew, "btcol")) return 1;
  }
  return 0;
}
