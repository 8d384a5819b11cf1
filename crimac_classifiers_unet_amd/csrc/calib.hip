// Measurement support: an MFMA-only calibration kernel.
//
// The roofline fractions of bench.py divide by the NOMINAL dense peak (2.5 PFLOP/s at 2.4 GHz).  Under a dense MFMA
// stream the chip holds a lower clock (power limit), and a box of the pool may run throttled for a whole pass --
// without an observation of the clock INSIDE the run a slow pass cannot be told from a slow kernel.  This kernel
// issues nothing but 16x16x32 bf16 MFMAs on register operands (no memory traffic, eight independent accumulators per
// wave, two waves per SIMD: the pipe is never idle) for a fixed count; its FLOP count over its HIP-event duration
// is the MFMA rate the chip delivers right now, and the s_memtime (shader cycles) / s_memrealtime (100 MHz) stamps
// of every workgroup give the shader clock directly.  bench.py runs it before and after the serialized per-kernel
// pass and prints both next to the roofline.  Not part of the product path.
#include "common.h"

__global__ __launch_bounds__(256) void mfma_calibrate_kernel(int iters, unsigned long long* stamps, float* sink) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (bf16_t)(0.001f * (float)((threadIdx.x + i) & 7));
    b[i] = (bf16_t)(0.002f * (float)((threadIdx.x * 3 + i) & 7));
  }
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, r0, t1, r1;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; ++it) {
    // in place on eight fixed accumulators (left to the compiler the loop carried accumulator shuffles around it)
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %8, %9, %3\n\t"
        "v_mfma_f32_16x16x32_bf16 %4, %8, %9, %4\n\t"
        "v_mfma_f32_16x16x32_bf16 %5, %8, %9, %5\n\t"
        "v_mfma_f32_16x16x32_bf16 %6, %8, %9, %6\n\t"
        "v_mfma_f32_16x16x32_bf16 %7, %8, %9, %7"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(a), "v"(b));
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678f) sink[0] = s;                  // (keeps the accumulators alive; never true)
  if (threadIdx.x == 0 && stamps != nullptr) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

extern "C" int crimac_mfma_calibrate(int iters, int blocks, unsigned long long* stamps, float* sink, void* stream) {
  CRIMAC_REQUIRE(iters > 0 && iters <= (1 << 24), "crimac_mfma_calibrate: iters %d out of range", iters);
  CRIMAC_REQUIRE(blocks > 0 && blocks <= 65536, "crimac_mfma_calibrate: blocks %d out of range", blocks);
  CRIMAC_REQUIRE(sink != nullptr, "crimac_mfma_calibrate: sink is NULL");
  mfma_calibrate_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, stamps, sink);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
