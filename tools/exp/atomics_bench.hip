// Micro-benchmark: how fast is the weight-gradient flush (fp32 atomics of one 9 x 64 x 64 accumulator set per workgroup)
// as a function of WHO shares an address -- all workgroups, the workgroups of one XCD, nobody?
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/exp/atomics_bench.hip -o gpurun_exp_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int TILE = 9 * 64 * 64;      // floats per workgroup
// mode 0: atomics, 1: plain stores.  dst tile of workgroup = base + tile_of[blockIdx.x] * TILE
__global__ __launch_bounds__(512) void flush(float* base, const int* tile_of, int mode, int by_xcc, int ntiles_per_slab) {
  int t = tile_of[blockIdx.x];
  if (by_xcc) t += (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) * ntiles_per_slab;
  float* dst = base + (long)t * TILE;
  const float v = 1.0f;
  if (mode >= 2) {
    // the weight-gradient kernels' flush shape: a wave-instruction = 4 rows x 16 floats (four 64-byte half lines, rows of
    // 64 floats); mode 3: the same elements as 2 rows x 32 floats (two whole 128-byte lines)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ws = wave & 1, row0 = (wave >> 1) * 144;             // a wave: one 32-column half of 144 of the 576 rows
    for (int k = 0; k < 72; ++k) {                                 // 72 wave-instructions either way
      int row, col;
      if (mode == 2) { row = row0 + (k >> 1) * 4 + (lane >> 4); col = ws * 32 + (k & 1) * 16 + (lane & 15); }
      else { row = row0 + k * 2 + (lane >> 5); col = ws * 32 + (lane & 31); }
      atomicAdd(dst + row * 64 + col, v);
    }
    return;
  }
  for (int i = threadIdx.x; i < TILE; i += 512) {
    if (mode == 0) atomicAdd(dst + i, v);
    else dst[i] = v;
  }
}
__global__ void xcc_of(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
int main() {
  const int NWG = 256;
  float* buf; CK(hipMalloc(&buf, (size_t)NWG * 8 * TILE * 4));
  int* d_tile; CK(hipMalloc(&d_tile, NWG * 4));
  int* d_x; CK(hipMalloc(&d_x, 1024 * 4));
  hipLaunchKernelGGL(xcc_of, dim3(1024), dim3(64), 0, 0, d_x);
  std::vector<int> xs(1024); CK(hipMemcpy(xs.data(), d_x, 4096, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += xs[i] != i % 8;
  printf("XCC_ID == blockIdx %% 8 for %d of 1024 workgroups\n", 1024 - bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, std::vector<int> tiles, int mode, int by_xcc, int per_slab) -> int {
    CK(hipMemcpy(d_tile, tiles.data(), NWG * 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf, 0, (size_t)NWG * 8 * TILE * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(flush, dim3(NWG), dim3(512), 0, 0, buf, d_tile, mode, by_xcc, per_slab);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(flush, dim3(NWG), dim3(512), 0, 0, buf, d_tile, mode, by_xcc, per_slab);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    printf("%-62s %7.1f us  %6.2f TB/s of 4-byte adds\n", name, us, (double)NWG * TILE * 4 / us / 1e6);
    return 0;
  };
  std::vector<int> t(NWG);
  for (int i = 0; i < NWG; ++i) t[i] = 0;
  run("atomics: all 256 workgroups -> ONE tile", t, 0, 0, 0);
  run("atomics: one tile per XCD (slab picked by XCC_ID)", t, 0, 1, 1);
  for (int i = 0; i < NWG; ++i) t[i] = i % 8;
  run("atomics: 8 tiles, sharers = the 32 workgroups with equal id % 8", t, 0, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = i / 32;
  run("atomics: 8 tiles, sharers = 32 consecutive ids (4 per XCD)", t, 0, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = (i % 8) * 8 + (i / 8) % 8;       // 64 tiles, 4 sharers, all on one XCD
  run("atomics: 64 tiles, 4 sharers on the SAME XCD", t, 0, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = i / 4;                           // 64 tiles, 4 sharers on 4 XCDs
  run("atomics: 64 tiles, 4 sharers on 4 XCDs", t, 0, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = i;
  run("atomics: private tiles (no sharing)", t, 0, 0, 0);
  run("plain stores: private tiles", t, 1, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = i / 4;
  run("atomics, 64 tiles x 4 sharers, 4 x 64-byte half lines per instruction", t, 2, 0, 0);
  run("atomics, 64 tiles x 4 sharers, 2 x 128-byte lines per instruction", t, 3, 0, 0);
  for (int i = 0; i < NWG; ++i) t[i] = 0;
  run("atomics, ONE tile, 4 x 64-byte half lines per instruction", t, 2, 0, 0);
  run("atomics, ONE tile, 2 x 128-byte lines per instruction", t, 3, 0, 0);
  return 0;
}
