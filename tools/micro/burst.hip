// Micro-benchmark: the load phase of an LDS-DMA convolution tile in isolation.  A workgroup of W wavefronts issues a
// burst of 32 KB (or 48 KB) of global_load_lds_dwordx4 (128-byte rows at a 512-byte stride, L2-resident data), waits
// for it (s_waitcnt vmcnt + barrier) and repeats; DEPTH bursts are kept in flight.  How long is one step?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// PER_WAVE instructions per wavefront and burst (1 KiB each); DEPTH = 1: wait for the burst just issued;
// DEPTH = 2: wait for the previous one while the newest is in flight
// SWZ: the 16-byte chunk a lane fetches is XOR-ed with its row (the swizzle the convolution tiles apply through the
// source address so that their fragment reads are bank-conflict free)
template <int WAVES, int PER_WAVE, int DEPTH, bool BARRIER = true, bool SWZ = false>
__global__ __launch_bounds__(WAVES * 64) void burst_kernel(const char* src, int steps, long long wg_stride_b,
                                                              int window_rows, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long long)blockIdx.x * wg_stride_b + (SWZ ? ((lane & 7) ^ (lane >> 3)) : (lane & 7)) * 16;
  const int lrow = lane >> 3;
  int row = wave * 8;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps + DEPTH - 1; ++s) {
    char* dst = smem + (s % DEPTH) * (WAVES * PER_WAVE * 1024) + wave * PER_WAVE * 1024;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      glds16(base + (long long)(row + lrow) * 512, dst + i * 1024);
      row += WAVES * 8;
      if (row >= window_rows) row -= window_rows;
    }
    if (DEPTH == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (PER_WAVE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    if (BARRIER) __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES, int PER_WAVE, int DEPTH, bool BARRIER = true, bool SWZ = false>
static void run(const char* name, const char* buf, int grid, long long wg_stride, int wgs_per_cu_note) {
  long long* cyc;
  (void)hipMalloc(&cyc, grid * sizeof(long long));
  const int steps = 2000, lds = DEPTH * WAVES * PER_WAVE * 1024;
  (void)hipFuncSetAttribute((const void*)burst_kernel<WAVES, PER_WAVE, DEPTH, BARRIER, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  burst_kernel<WAVES, PER_WAVE, DEPTH, BARRIER, SWZ><<<grid, WAVES * 64, lds>>>(buf, 50, wg_stride, 4096, cyc);
  (void)hipEventRecord(e0);
  burst_kernel<WAVES, PER_WAVE, DEPTH, BARRIER, SWZ><<<grid, WAVES * 64, lds>>>(buf, steps, wg_stride, 4096, cyc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid);
  (void)hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= grid;
  const double kb = WAVES * PER_WAVE;
  printf("%-34s grid %4d  %2d KB/burst  depth %d  %7.1f ns/step  %6.0f cycles/step  %6.1f B/clk/WG  %6.2f TB/s total\n", name,
         grid, (int)kb, DEPTH, ms * 1e6 / steps, avg / steps, kb * 1024 / (avg / steps), kb * 1024.0 * grid / (ms * 1e6 / steps) / 1e3);
  (void)hipFree(cyc);
}

int main() {
  char* buf;
  const long long bytes = 512ll << 20;
  (void)hipMalloc(&buf, bytes);
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  // every workgroup its own 2 MB window (4096 rows x 512 B): activations-like, L2/MALL resident after the warm-up
  for (int grid : {1, 64, 256}) {
    run<4, 8, 1>("4 waves x 8, wait all", buf, grid, 1 << 20, 1);
    run<4, 8, 2>("4 waves x 8, one burst ahead", buf, grid, 1 << 20, 1);
    run<8, 4, 1>("8 waves x 4, wait all", buf, grid, 1 << 20, 1);
    run<8, 4, 2>("8 waves x 4, one burst ahead", buf, grid, 1 << 20, 1);
    run<4, 12, 1>("4 waves x 12, wait all", buf, grid, 1 << 20, 1);
    run<4, 4, 1>("4 waves x 4 (16 KB), wait all", buf, grid, 1 << 20, 1);
  }
  // no barrier (waves drift apart), and two / three workgroups per CU
  run<4, 8, 1, false>("shared, 4 x 8, wait all, no barrier", buf, 256, 0, 1);
  run<4, 8, 2, false>("shared, 4 x 8, one ahead, no barrier", buf, 256, 0, 1);
  run<8, 4, 2, false>("shared, 8 x 4, one ahead, no barrier", buf, 256, 0, 1);
  run<4, 8, 1>("shared, 4 x 8, wait all, 2 WG/CU", buf, 512, 0, 2);
  run<4, 8, 1>("shared, 4 x 8, wait all, 3 WG/CU", buf, 768, 0, 3);
  run<4, 8, 2>("shared, 4 x 8, one ahead, 2 WG/CU", buf, 512, 0, 2);
  run<4, 8, 1, true, true>("shared, 4 x 8, wait all, 2 WG/CU, SWIZZLED", buf, 512, 0, 2);
  run<4, 8, 1, true, true>("shared, 4 x 8, wait all, 1 WG/CU, SWIZZLED", buf, 256, 0, 1);
  run<4, 12, 1, true, true>("shared, 4 x 12, wait all, 2 WG/CU, SWIZZLED", buf, 512, 0, 2);
  run<4, 12, 1, true, false>("shared, 4 x 12, wait all, 2 WG/CU", buf, 512, 0, 2);
  // shared window (weights-like): all workgroups read the same 2 MB
  run<4, 8, 1>("shared window: 4 x 8, wait all", buf, 256, 0, 1);
  run<4, 8, 2>("shared window: 4 x 8, one ahead", buf, 256, 0, 1);
  return 0;
}
