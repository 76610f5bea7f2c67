// How fast does the chip absorb the projection GEMMs' output write, by store shape?  A (M x N) bf16 matrix is written once by
// 256-row x 256-column "tiles" (one 512-thread workgroup each, like csrc/linear_tiled.hip), each wave owning 64 rows x 128
// columns (256 B per row), with wave instructions that cover  8 rows x 128 B | 4 rows x 256 B ; or each wave owning 32 rows x
// all 256 columns with 2 rows x 512 B per instruction; against a plain linear fill.  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ void __launch_bounds__(512) tile_store(unsigned short* y, int M, int N, int ntn) {
  const int tile = blockIdx.x, tm = tile / ntn, tn = tile % ntn;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, (unsigned)tile};
  if (MODE == 0) {          // 8 rows x 128 B per instruction, two 128-B column halves one after the other (the shipped epilogue)
    const int wm = wid >> 1, wn = wid & 1;
    for (int nh = 0; nh < 2; ++nh)
      for (int j = 0; j < 8; ++j) {
        const int row = tm * 256 + wm * 64 + 8 * j + (lane >> 3);
        *(u32x4*)(y + (size_t)row * N + tn * 256 + wn * 128 + nh * 64 + (lane & 7) * 8) = v;
      }
  } else if (MODE == 1) {   // 4 rows x 256 B per instruction
    const int wm = wid >> 1, wn = wid & 1;
    for (int j = 0; j < 16; ++j) {
      const int row = tm * 256 + wm * 64 + 4 * j + (lane >> 4);
      *(u32x4*)(y + (size_t)row * N + tn * 256 + wn * 128 + (lane & 15) * 8) = v;
    }
  } else {                  // 2 rows x 512 B per instruction: a wave owns 32 whole tile rows
    for (int j = 0; j < 16; ++j) {
      const int row = tm * 256 + wid * 32 + 2 * j + (lane >> 5);
      *(u32x4*)(y + (size_t)row * N + tn * 256 + (lane & 31) * 8) = v;
    }
  }
}
__global__ void linear_fill(u32x4* y, size_t n16) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) y[i] = v;
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 3840;
  unsigned short* y;
  const size_t bytes = (size_t)M * N * 2;
  hipMalloc(&y, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ntn = N / 256, tiles = (M / 256) * ntn;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %7.1f us  %5.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  run("linear fill, 16 B per lane", [&] { hipLaunchKernelGGL(linear_fill, dim3(2048), dim3(256), 0, 0, (u32x4*)y, bytes / 16); });
  run("tiles, 8 rows x 128 B per instruction", [&] { hipLaunchKernelGGL((tile_store<0>), dim3(tiles), dim3(512), 0, 0, y, M, N, ntn); });
  run("tiles, 4 rows x 256 B per instruction", [&] { hipLaunchKernelGGL((tile_store<1>), dim3(tiles), dim3(512), 0, 0, y, M, N, ntn); });
  run("tiles, 2 rows x 512 B per instruction", [&] { hipLaunchKernelGGL((tile_store<2>), dim3(tiles), dim3(512), 0, 0, y, M, N, ntn); });
  return 0;
}
