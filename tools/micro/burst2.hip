// Micro-benchmark: what limits the LDS-DMA load phase of a convolution tile - the issue of the instructions, their
// latency, or the waits a __syncthreads() implies?  A workgroup of W wavefronts issues bursts of 1-KiB loads into LDS
// (128-byte rows at a 512-byte stride, L2-resident), DEPTH bursts in flight, and synchronises per burst with
// __syncthreads() (which the compiler precedes with s_waitcnt vmcnt(0)), a bare s_barrier, or not at all.
// MODE 0: global_load_lds_dwordx4 (64-bit per-lane addresses); MODE 1: buffer_load_dwordx4 ... lds (resource +
// 32-bit per-lane offset + scalar offset).  Also reports the cycles a wavefront spends issuing one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int WAVES, int PER_WAVE, int DEPTH, int BAR, int MODE>
__global__ __launch_bounds__(WAVES * 64) void burst_kernel(const char* src, int steps, long long wg_stride_b,
                                                              int window_rows, long long* cycles, long long* issue_cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* wg = src + (long long)blockIdx.x * wg_stride_b;
  const unsigned lane_off = (lane >> 3) * 512 + (lane & 7) * 16;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, 0x7fffffff, 0x00020000);
  int row = wave * 8;
  long long issue = 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps + DEPTH - 1; ++s) {
    char* dst = smem + (s % DEPTH) * (WAVES * PER_WAVE * 1024) + wave * PER_WAVE * 1024;
    const long long i0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      if (MODE == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (long long)row * 512 + lane_off),
                                         (lds_ptr)(dst + i * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(dst + i * 1024), 16, lane_off, row * 512, 0, 0);
      row += WAVES * 8;
      if (row >= window_rows) row -= window_rows;
    }
    issue += __builtin_readcyclecounter() - i0;
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) {
      if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (PER_WAVE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (BAR == 0) __syncthreads();
    if (BAR == 1) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    cycles[blockIdx.x] = t1 - t0;
    issue_cycles[blockIdx.x] = issue;
  }
}

template <int WAVES, int PER_WAVE, int DEPTH, int BAR, int MODE>
static void run(const char* name, const char* buf, int grid, long long wg_stride) {
  long long *cyc, *iss;
  (void)hipMalloc(&cyc, grid * sizeof(long long));
  (void)hipMalloc(&iss, grid * sizeof(long long));
  const int steps = 2000, lds = DEPTH * WAVES * PER_WAVE * 1024;
  auto kern = burst_kernel<WAVES, PER_WAVE, DEPTH, BAR, MODE>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kern<<<grid, WAVES * 64, lds>>>(buf, 50, wg_stride, 4096, cyc, iss);
  (void)hipEventRecord(e0);
  kern<<<grid, WAVES * 64, lds>>>(buf, steps, wg_stride, 4096, cyc, iss);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid), hi(grid);
  (void)hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hi.data(), iss, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0, avgi = 0;
  for (auto c : h) avg += c;
  for (auto c : hi) avgi += c;
  avg /= grid;
  avgi /= grid;
  const double kb = WAVES * PER_WAVE;
  printf("%-14s %dx%-2d depth %d %-12s grid %4d  %7.1f ns/step %6.0f cyc/step %6.1f B/clk/WG  issue %5.1f cyc/instr\n", name, WAVES,
         PER_WAVE, DEPTH, BAR == 0 ? "syncthreads" : BAR == 1 ? "s_barrier" : "no barrier", grid, ms * 1e6 / steps, avg / steps,
         kb * 1024 / (avg / steps), avgi / (steps + DEPTH - 1) / PER_WAVE);
  (void)hipFree(cyc);
  (void)hipFree(iss);
}


// After issuing a burst into one LDS stage: how long until a ds_read_b128 of the OTHER stage returns to the same
// wavefront (s_waitcnt lgkmcnt(0) alone, the burst still in flight)?
template <int PER_WAVE>
__global__ __launch_bounds__(256) void probe_kernel(const char* src, int steps, long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane_off = (lane >> 3) * 512 + (lane & 7) * 16;
  int row = wave * 8;
  long long t_read = 0, t_all = 0;
  unsigned sink = 0;
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    char* dst = smem + (s & 1) * (4 * PER_WAVE * 1024) + wave * PER_WAVE * 1024;
    const char* other = smem + ((s + 1) & 1) * (4 * PER_WAVE * 1024) + threadIdx.x * 16;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)row * 512 + lane_off),
                                       (lds_ptr)(dst + i * 1024), 16, 0, 0);
      row += 32;
      if (row >= 4096) row -= 4096;
    }
    const long long t1 = __builtin_readcyclecounter();
    const uint4 v = *reinterpret_cast<const uint4*>(other);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    sink ^= v.x ^ v.y ^ v.z ^ v.w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    t_read += t2 - t1;
    t_all += t3 - t0;
  }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 3 + 0] = t_read;
    out[blockIdx.x * 3 + 1] = t_all;
    out[blockIdx.x * 3 + 2] = sink;
  }
}

template <int PER_WAVE>
static void probe(const char* buf) {
  long long* out;
  const int grid = 256, steps = 2000, lds = 2 * 4 * PER_WAVE * 1024;
  (void)hipMalloc(&out, grid * 3 * sizeof(long long));
  (void)hipFuncSetAttribute((const void*)probe_kernel<PER_WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  probe_kernel<PER_WAVE><<<grid, 256, lds>>>(buf, 50, out);
  probe_kernel<PER_WAVE><<<grid, 256, lds>>>(buf, steps, out);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(grid * 3);
  (void)hipMemcpy(h.data(), out, grid * 3 * sizeof(long long), hipMemcpyDeviceToHost);
  double r = 0, a = 0;
  for (int i = 0; i < grid; ++i) {
    r += h[i * 3];
    a += h[i * 3 + 1];
  }
  printf("probe: %d loads per wavefront in flight: ds_read_b128 of the other stage returns after %.0f cycles; the burst lands after %.0f\n",
         PER_WAVE, r / grid / steps, a / grid / steps);
  (void)hipFree(out);
}

template <int MODE>
static void all(const char* name, const char* buf) {
  for (int grid : {256, 512}) {
    run<4, 8, 1, 0, MODE>(name, buf, grid, 0);
    run<4, 8, 1, 1, MODE>(name, buf, grid, 0);
    run<4, 8, 2, 0, MODE>(name, buf, grid, 0);
    run<4, 8, 2, 1, MODE>(name, buf, grid, 0);
    run<4, 8, 3, 1, MODE>(name, buf, grid, 0);
    run<8, 4, 2, 1, MODE>(name, buf, grid, 0);
    run<8, 4, 3, 1, MODE>(name, buf, grid, 0);
    run<4, 12, 2, 1, MODE>(name, buf, grid, 0);
    run<4, 8, 2, 2, MODE>(name, buf, grid, 0);
  }
}

int main() {
  char* buf;
  const long long bytes = 512ll << 20;
  (void)hipMalloc(&buf, bytes);
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  probe<8>(buf);
  probe<4>(buf);
  probe<1>(buf);
  all<0>("global_load_lds", buf);
  all<1>("buffer_load_lds", buf);
  return 0;
}
