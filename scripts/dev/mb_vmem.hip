// Microbenchmark (dev tool, round 5): what a wavefront pays to ISSUE the staging requests and the stores of k_mass_flux_wave, as a
// function of how many distinct 128-byte lines one instruction touches.  hipcc --offload-arch=gfx950 -O3 -o mb_vmem mb_vmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const double *src, double *lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// MODE 0: quarter lines (32 B of 32 lines per instruction, the kernel's pattern), 15 instructions per wavefront and row
// MODE 1: full lines (8 lines per instruction), the same bytes per work-group: 3 arrays x 75 lines = 225 lines = 29 instructions per WG
// STORE 0: none; 1: the kernel's pattern (8 B per lane, 16 lines x 32 B per instruction), 10 per row; 2: full lines (4 lines per instruction)
template <int MODE, int STORE>
__global__ void __launch_bounds__(256, 2) k(const double *A, double *B, size_t slab, int pitch, int gx, int rows, int nk, int spin,
                                           unsigned long long *out) {
  extern __shared__ double S[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bx = blockIdx.x % gx, chunk = blockIdx.x / gx;
  const int i0 = bx * 16, j0 = chunk * rows;
  double *Sw = S + w * 1700;
  unsigned long long t_issue = 0, t_store = 0, t_wait = 0;
  double acc = (double)tid;
  for (int jj = j0; jj < j0 + rows; jj++) {
    const size_t row = (size_t)jj * pitch + i0;
    long long t0 = clock64();
    if (MODE == 0) {
      const int sg = lane >> 1, pp = lane & 1;
#pragma unroll
      for (int s = 0; s < 5; s++)
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (r * 32 + sg < nk) glds16(A + (size_t)(s % 3) * 80 * slab + (size_t)(r * 32 + sg) * slab + row + (s / 3) * 4 + w * 4 + pp * 2, Sw + (s * 96 + r * 32) * 4);
    } else {
      const int kk = lane >> 3, piece = lane & 7;
      for (int q = w; q < 29; q += 4) {
        const int arr = q / 10, m = q % 10;
        if (m * 8 + kk < nk) glds16(A + (size_t)arr * 80 * slab + (size_t)(m * 8 + kk) * slab + row + piece * 2, S + q * 128);
      }
    }
    long long t1 = clock64();
    t_issue += (unsigned long long)(t1 - t0);
    // the row's arithmetic
    for (int q = 0; q < spin; q++) acc = acc * 1.0000001 + 0.5;
    long long t2 = clock64();
    if (STORE == 1) {
      const int fw = lane >> 4, kl = lane & 15;
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int n = 0; n < 5; n++)
          if (kl + 16 * n < nk) B[(size_t)a * 80 * slab + (size_t)(kl + 16 * n) * slab + row + w * 4 + fw] = acc;
    } else if (STORE == 2) {
      const int kk = lane >> 4, c = lane & 15;
      for (int q = w; q < 38; q += 4) {
        const int a = q / 19, m = q % 19;
        if (m * 4 + kk < nk) B[(size_t)a * 80 * slab + (size_t)(m * 4 + kk) * slab + row + c] = acc;
      }
    }
    long long t3 = clock64();
    t_store += (unsigned long long)(t3 - t2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long long t4 = clock64();
    t_wait += (unsigned long long)(t4 - t3);
  }
  if (acc == 12345.678) out[7] = 1;
  if (lane == 0) { atomicAdd(&out[0], t_issue); atomicAdd(&out[1], t_store); atomicAdd(&out[2], t_wait); atomicAdd(&out[3], 1ull); }
}

template <int MODE, int STORE>
void run(const char *name, const double *A, double *B, size_t slab, int pitch, int gx, int gy, int rows, int nk, int spin, unsigned long long *out) {
  CHK(hipMemset(out, 0, 64));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const size_t lds = 4 * 1700 * 8;
  for (int rep = 0; rep < 2; rep++) {
    CHK(hipMemset(out, 0, 64));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, STORE>), dim3(gx * gy), dim3(256), lds, 0, A, B, slab, pitch, gx, rows, nk, spin, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  }
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8]; CHK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
  const double per = 1.0 / ((double)h[3] * rows);
  printf("%-34s %7.3f ms   per wavefront-row: issue %7.0f cyc, stores %7.0f cyc, wait+barrier %7.0f cyc\n", name, ms, h[0] * per, h[1] * per, h[2] * per);
}

int main(int argc, char **argv) {
  const int ni = 1440, nj = 1080, nk = 75, pitch = 1456, spin = argc > 1 ? atoi(argv[1]) : 1500;
  const size_t slab = (size_t)pitch * (nj + 9);
  double *A, *B; unsigned long long *out;
  CHK(hipMalloc(&A, 3 * 80 * slab * 8)); CHK(hipMalloc(&B, 2 * 80 * slab * 8)); CHK(hipMalloc(&out, 64));
  CHK(hipMemset(A, 0, 3 * 80 * slab * 8)); CHK(hipMemset(B, 0, 2 * 80 * slab * 8));
  const int gx = ni / 16, rows = 16, gy = (nj + rows - 1) / rows;
  printf("spin = %d dependent FMAs per row\n", spin);
  run<0, 0>("quarter-line DMA, no stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  run<1, 0>("full-line DMA, no stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  run<0, 1>("quarter-line DMA, 32-B stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  run<1, 1>("full-line DMA, 32-B stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  run<0, 2>("quarter-line DMA, full-line stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  run<1, 2>("full-line DMA, full-line stores", A, B, slab, pitch, gx, gy, rows, nk, spin, out);
  return 0;
}
