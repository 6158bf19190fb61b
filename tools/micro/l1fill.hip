// Micro-benchmark: how many bytes per cycle and CU does LDS-DMA (global_load_lds_dwordx4) deliver from L2, as a
// function of the contiguous run per row (64 B / 128 B / 1 KiB) and of the row stride?  One 512-thread workgroup per
// CU; every wavefront streams `iters` instructions over a window that is larger than the 32 KB vector L1 and far
// smaller than L2, so every line comes from L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// run_b: contiguous bytes per row (64, 128, 1024); stride_b: bytes between rows; window_rows: rows in the window
template <int RUN_B>
__global__ __launch_bounds__(512, 1) void fill_kernel(const char* src, long long stride_b, int window_rows, int iters,
                                                      long long wg_offset_b, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = RUN_B / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per instruction
  const char* base = src + (long long)blockIdx.x * wg_offset_b + (long long)(lane % LPR) * 16;
  const int lrow = lane / LPR;
  char* dst = smem + wave * 8192;
  int row = wave * RPI;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      glds16(base + (long long)(row + lrow) * stride_b, dst + u * 1024);
      row += 8 * RPI;
      if (row >= window_rows) row -= window_rows;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int RUN_B>
static void run(const char* name, const char* buf, long long stride_b, int window_rows, long long wg_off, int grid) {
  long long* cyc;
  hipMalloc(&cyc, grid * sizeof(long long));
  const int iters = 4096;
  hipFuncSetAttribute((const void*)fill_kernel<RUN_B>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  fill_kernel<RUN_B><<<grid, 512, 65536>>>(buf, stride_b, window_rows, 64, wg_off, cyc);
  hipEventRecord(e0);
  fill_kernel<RUN_B><<<grid, 512, 65536>>>(buf, stride_b, window_rows, iters, wg_off, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= grid;
  const double bytes_wg = 8.0 * iters * 1024;
  printf("%-44s grid %3d  %8.3f ms  %7.2f B/clk/CU (s_memtime @100MHz: x24 at 2.4 GHz)  %7.2f GB/s/CU  total %6.2f TB/s\n",
         name, grid, ms, bytes_wg / avg, bytes_wg / (ms * 1e-3) / 1e9, bytes_wg * grid / (ms * 1e-3) / 1e12);
  hipFree(cyc);
}

int main() {
  char* buf;
  const long long bytes = 256ll << 20;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int grid : {1, 32, 256}) {
    // window: 2048 rows; all workgroups read the same window (weights-like, L2 resident)
    run<1024>("1 KiB runs, contiguous, shared window", buf, 1024, 512, 0, grid);
    run<128>("128 B rows, stride 128 (contig), shared", buf, 128, 4096, 0, grid);
    run<128>("128 B rows, stride 512, shared", buf, 512, 1024, 0, grid);
    run<64>("64 B rows, stride 512, shared", buf, 512, 1024, 0, grid);
    run<64>("64 B rows, stride 64 (contig), shared", buf, 64, 8192, 0, grid);
    run<128>("128 B rows, stride 4096, shared", buf, 4096, 1024, 0, grid);
    run<64>("64 B rows, stride 4096, shared", buf, 4096, 1024, 0, grid);
    // private windows (activation-like): 512 KB per workgroup
    run<128>("128 B rows, stride 512, private 512 KB", buf, 512, 1024, 512 << 10, grid);
    run<64>("64 B rows, stride 512, private 512 KB", buf, 512, 1024, 512 << 10, grid);
    // L1-resident window (16 KB): the TA / LDS-write limit without L2
    run<128>("128 B rows, stride 128, 16 KB window (L1)", buf, 128, 128, 0, grid);
    run<64>("64 B rows, stride 512, 16 KB window (L1)", buf, 512, 256 / 8 * 8, 0, grid);
  }
  return 0;
}
