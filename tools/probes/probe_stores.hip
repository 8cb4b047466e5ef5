// Store-path probe (round 4): how fast can ONE CU / the whole chip push a GEMM tile's epilogue stores, by the shape of a
// store instruction?  A wave stores 1 KB per instruction (64 lanes x 16 B) as SEG-byte contiguous row segments of an output with
// row stride LD bytes; 8 waves per workgroup, one workgroup per CU, `grid` workgroups, each writing `reps` tiles of 128 KB.
//   hipcc --offload-arch=gfx950 -O3 -o probe_stores tools/probes/probe_stores.hip && ./probe_stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int SEG>
__global__ void __launch_bounds__(512) store_kernel(char* out, long ld, int reps, long tile_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPS = SEG / 16;   // lanes per segment
  constexpr int RPI = 64 / LPS;   // rows per instruction
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  for (int r = 0; r < reps; ++r) {
    const long tidx = (long)r * gridDim.x + blockIdx.x;
    // tile_stride 0: tiles are 256 x 512-B windows of a [rows, ld] matrix, 12 side by side (a GEMM output with N = 3072)
    char* tile = tile_stride ? out + tidx * tile_stride : out + (tidx / 12) * 256 * ld + (tidx % 12) * 512;
    // a 256 x 256 bf16 tile = 256 rows x 512 B; wave w owns rows 32 w .. 32 w + 31 = 16 KB = 16 instructions
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long unit = (long)i * 64 + lane;            // 16-byte unit index inside the wave's 16 KB
      const long seg = unit / LPS, within = unit % LPS;   // segment index, position inside it
      const long segs_per_row = 512 / SEG;
      const long row = 32 * wave + seg / segs_per_row, col = (seg % segs_per_row) * SEG + within * 16;
      *reinterpret_cast<u4*>(tile + row * ld + col) = v;
    }
  }
}

template <int SEG>
void run(char* buf, int grid, int reps, long ld, long tile_stride, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  store_kernel<SEG><<<grid, 512>>>(buf, ld, reps, tile_stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) store_kernel<SEG><<<grid, 512>>>(buf, ld, reps, tile_stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5, bytes = (double)grid * reps * 131072;
  printf("%-10s seg %4d B  grid %3d reps %3d: %8.1f us  %7.1f GB/s per CU  %6.2f TB/s chip\n", what, SEG, grid, reps, us,
         bytes / grid / us / 1e3, bytes / us / 1e6);
}

int main() {
  const long total = 1L << 30;
  char* buf;
  hipMalloc(&buf, total + (64 << 20));
  hipMemset(buf, 0, total);
  // (a) a GEMM output: row stride 6144 B (N = 3072 bf16); tiles side by side -> tile_stride = 512 B along the row... use row blocks:
  // tile t at byte offset (t / 12) * 256 * 6144 + (t % 12) * 512
  for (int grid : {256, 128, 64, 32}) {
    const int reps = 12;
    // tiles laid out as disjoint 256-row x 512-B windows of a [M, 3072] bf16 matrix is awkward with a single stride; use
    // dense windows instead: each tile owns 256 rows x LD bytes with LD = 6144 and writes the first 512 B of each row
    const long ld = 6144, ts = 0;
    run<64>(buf, grid, reps, ld, ts, "strided");
    run<128>(buf, grid, reps, ld, ts, "strided");
    run<256>(buf, grid, reps, ld, ts, "strided");
    run<512>(buf, grid, reps, ld, ts, "strided");
    // (b) fully dense tiles (128 KB contiguous)
    run<64>(buf, grid, reps, 512, 131072, "dense");
    run<512>(buf, grid, reps, 512, 131072, "dense");
  }
  return 0;
}
