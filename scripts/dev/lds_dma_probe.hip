// Dev probe: does LDS-DMA (global_load_lds_dwordx4) reach LDS addresses beyond 64 KB on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* __restrict__ src, double* dst, int line_off) {
  extern __shared__ double sm[];
  const int t = threadIdx.x;
  for (int q = t; q < 20480; q += 256) sm[q] = -1.0;
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const double* g = src + t * 2;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(sm + (size_t)line_off * 16 + w * 128), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  dst[t] = sm[(size_t)line_off * 16 + t];
  dst[256 + t] = sm[(size_t)line_off * 16 + 256 + t];
  // where did it land if wrapped?  report the first LDS double that equals src[0]
  if (t == 0) { int found = -1; for (int q = 0; q < 20480; q++) if (sm[q] == 1000.0) { found = q; break; } dst[512] = found; }
}
int main() {
  double *s, *d; hipMalloc(&s, 512 * 8); hipMalloc(&d, 520 * 8);
  double h[520]; for (int i = 0; i < 512; i++) h[i] = 1000.0 + i;
  hipMemcpy(s, h, 512 * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int off : {0, 500, 520, 600, 1000, 1200}) {   // lines of 128 B: 512 lines = 64 KB
    hipMemset(d, 0, 520 * 8);
    k<<<1, 256, 163840>>>(s, d, off);
    hipMemcpy(h, d, 520 * 8, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 512; i++) if (h[i] != 1000.0 + i) ok = 0;
    printf("line offset %4d (byte %7d): %s; first copy found at double index %d (expected %d)\n", off, off * 128, ok ? "OK" : "WRONG", (int)h[512], off * 16);
  }
  return 0;
}
