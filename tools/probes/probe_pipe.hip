// Vector-memory pipe probe (round 4): does a CU's LDS-DMA load stream (L2 hits) slow down out of proportion when stores are mixed in?
// 8 waves per workgroup, one workgroup per CU.  Per iteration a wave issues NL LDS-DMA loads of 1 KB (global_load_lds_dwordx4 from
// a 2 MB region that every workgroup re-reads: L2 / MALL hits) and, every `every` iterations, NS stores of 1 KB to its own slice
// of a large output; a counted vmcnt keeps ~DEPTH operations in flight per wave.
//   hipcc -w --offload-arch=gfx950 -O3 -o probe_pipe tools/probes/probe_pipe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NL, int NS>
__global__ void __launch_bounds__(512) pipe_kernel(const char* src, char* out, int iters, int every, long out_per_wg) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  char* lds = smem + wave * 16384;  // 16 slots of 1 KB per wave
  const char* s = src + (wave * 64 + lane) * 16;
  char* o = out + (long)blockIdx.x * out_per_wg + (wave * 64 + lane) * 16;
  long soff = 0, ooff = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + soff),
                                       (__attribute__((address_space(3))) void*)(lds + ((it * NL + i) & 15) * 1024), 16, 0, 0);
      soff = (soff + 8192) & (2097152 - 1);
    }
    if (NS > 0 && it % every == 0) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        *reinterpret_cast<u4*>(o + ooff) = v;
        ooff += 8192;
        if (ooff >= out_per_wg) ooff = 0;
      }
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  }
}

template <int NL, int NS>
void run(const char* src, char* out, int grid, int iters, int every, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const long per = 4L << 20;
  pipe_kernel<NL, NS><<<grid, 512, 131072>>>(src, out, iters, every, per);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) pipe_kernel<NL, NS><<<grid, 512, 131072>>>(src, out, iters, every, per);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 3;
  const double lb = 8.0 * iters * NL * 1024, sb = NS ? 8.0 * ((iters + every - 1) / every) * NS * 1024 : 0;
  printf("%-26s grid %3d: %8.1f us  loads %6.1f GB/s per CU  stores %6.1f GB/s per CU  (%.0f KB + %.0f KB per CU)\n", what, grid, us,
         lb / us / 1e3, sb / us / 1e3, lb / 1024, sb / 1024);
}

int main() {
  char *src, *out;
  hipMalloc(&src, 4 << 20);
  hipMalloc(&out, 256L * (4L << 20));
  hipMemset(src, 1, 4 << 20);
  hipFuncSetAttribute((const void*)pipe_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)pipe_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)pipe_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)pipe_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int grid : {256, 64}) {
    run<2, 0>(src, out, grid, 600, 1, "loads only");                       // 1200 KB per wave... 9.6 MB per CU
    run<0, 2>(src, out, grid, 100, 1, "stores only");                      // 200 KB per wave = 1.6 MB per CU
    run<2, 2>(src, out, grid, 600, 6, "loads + stores 6:1 bursts of 2");   // every 6th iteration 2 stores: 6:1 bytes
    run<2, 1>(src, out, grid, 600, 3, "loads + stores 6:1 single");        // every 3rd iteration 1 store
    run<2, 2>(src, out, grid, 600, 2, "loads + stores 2:1");
  }
  return 0;
}
