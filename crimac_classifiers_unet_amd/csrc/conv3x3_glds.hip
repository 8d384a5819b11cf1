// 3x3 convolution (stride 1, pad 1), bf16, LDS-DMA kernels for CDNA4 (gfx950).
//
// Same contraction and epilogue as conv3x3.hip (reference: nn.Conv2d 3x3 forward and input gradient,
// unet.py:35-44, pipeline.py:177), restructured after measuring the register-staged kernel
// (tools/bench_conv.py ablations, 1024->1024 @16x16, B=32): the VGPR->LDS stores of the staged tiles cost
// 36 % of its time, the loads 24 %, the MFMAs only ~45 %.  Common to both kernels here:
//   * 16x16-pixel tile, 4 waves, two workgroups per CU (<= 256 registers, <= 74 KB LDS);
//   * the halo (18x18 pixels x 64 channels) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4): no
//     staging registers, no ds_write; the XOR swizzle is applied on the per-lane SOURCE address (the LDS
//     image of one wave-instruction is lane-linear: 8 rows x 128 B), and it is read 9x at shifted rows
//     (im2col-free);
//   * halo rows outside the image never change for a workgroup: zeroed once, those lanes masked off.
// conv3x3_wch_kernel  (N % 128 == 0): waves split the output CHANNELS, weights go global -> registers,
//                     no barrier inside a 64-channel chunk, inline-asm read/MFMA pipeline, 16x16x32 MFMAs
//                     (1.1-1.6 PFLOP/s);
// conv3x3_glds_w4_kernel (N = 64):    waves split the PIXELS, weights through a 3-slot LDS ring with counted
//                     vmcnt + raw s_barrier per tap (the HBM-heavy 256x256 layers).
// An 8-wave / one-workgroup-per-CU form with a double-buffered halo was the first LDS-DMA kernel; the 4-wave
// forms beat it by 10-20 % (wch: 20-35 %) on every layer and it was removed.
#include <stdlib.h>


#include "common.h"
#include "conv_epilogue.h"

CRIMAC_DIAG_DECLARE(crimac_diag_clock_conv)
#ifdef CRIMAC_DIAG_EPI
extern "C" int crimac_diag_epi_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(crimac_epi_buf), sizeof(unsigned long long) * 1024 * 8);
}
#endif
#ifndef CRIMAC_P64_HALO_AUX
#define CRIMAC_P64_HALO_AUX 0      // cache policy of the halo loads (2: non-temporal)
#endif
#ifndef CRIMAC_W4_NSLOT64
#define CRIMAC_W4_NSLOT64 3        // weight-ring slots of the 64-channel pixel-split kernel (the ring runs NSLOT - 1 taps ahead)
#endif
#ifndef CRIMAC_WCH_HALO_AUX
#define CRIMAC_WCH_HALO_AUX 0
#endif

namespace {

struct ConvParams {
  const void* in;
  long in_ld;
  int B, H, W;
  int Cin, N;
  const unsigned short* w_hi;
  EpiParams epi;
  int tiles_y, tiles_x;
  int n_first, n_count;      // output-channel range of this launch (crimac_conv3x3_cols); default the whole N
  int wfrag = 0;             // the weight plane is FRAGMENT-MAJOR (CRIMAC_EPI_WFRAG, common.h wfrag_index): channel-split kernel only
};

// Epilogue staging of an output storage type: 16-bit types stage the whole 256-row tile; fp32 and plane pairs (staged
// as fp32) go through the staging area in two slices of 128 rows (conv_epilogue.h) so that two workgroups still fit a CU.
template <typename TO> struct EpiPasses {
  static constexpr int kStageBytes = sizeof(TO) == 2 ? 2 : 4;
  static constexpr int value = sizeof(TO) == 2 ? 1 : 2;
};

constexpr int TR = 16, TC = 16, HP = TC + 2;
constexpr int BM = TR * TC;                       // 256 pixels
constexpr int HALO_ROWS = (TR + 2) * HP;          // 324
constexpr int HALO_INSTR = (HALO_ROWS + 7) / 8;   // 41 wave-instructions of 8 rows
constexpr int BK = 64, RB = BK * 2;               // 64 channels = 128 B per row
constexpr int A_BYTES = HALO_INSTR * 1024;        // one halo buffer (padded to whole instructions)

// XOR swizzle of the 16-byte unit index of halo column hx (0..17).  A 16x16x32 A fragment is read by
// ds_read_b128 with lane -> (column fr + kx, unit 4*ks + lane / 16); the hardware serves it in four groups of
// 16 lanes that mix two k units ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS table), so the plain
// (hx >> 1) & 7 swizzle of the 32x32x16 layout is 2-way conflicted for kx = 1, 2 (8 LDS cycles instead of 4,
// by the model).  This table (exhaustive search) is conflict-free for kx = 0, 1, 2: 3 bits per column.
// rocprofv3 PMC, 1024->512 @32x32: SQ_LDS_BANK_CONFLICT 126.5 M -> 0.66 M cycles, SQ_LDS_IDX_ACTIVE 321 M ->
// 195 M; run time unchanged (196 us either way: the MFMA stream, not the LDS, bounds the kernel) -- kept for
// the idle LDS cycles it leaves to the halo DMA.  -DCRIMAC_EXP_OLDSWZ restores the plain swizzle for A/B runs.
__device__ __forceinline__ int halo_swz(int hx) {
#ifdef CRIMAC_EXP_OLDSWZ
  return (hx >> 1) & 7;
#else
  return (int)((0xd92dad912240ull >> (3 * hx)) & 7);
#endif
}

// Plane-pair input (hp_t, common.h): a 32-channel chunk of a pixel is 128 bytes = 4 groups of [8 hi | 8 lo] halves.
// The LDS image keeps the 16-bit kernels' shape -- logical 16-byte units 0-3 = the first k-step of 32 (here: the hi
// plane of the 32 channels), units 4-7 = the second (the lo plane) -- so logical unit u comes from source unit
// 2 * (u & 3) + (u >> 2); the permutation rides on the per-lane DMA source address like the bank swizzle does.
template <bool PP> __device__ __forceinline__ int src_unit(int u) { return PP ? (((u & 3) << 1) | (u >> 2)) : u; }

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
#ifdef CRIMAC_EXP_NOGLDS
  return;
#endif
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  static_assert(N == 0 || N == 1 || N == 2 || N == 4, "add the immediate");
}

// ---------------------------------------------------------------------------------------------------
// N = 64 layers (level 0 / decoder 3: 256x256 images, Cin 64 or 128).  With the 8-wave kernel above a
// 64-channel tile leaves each wave a 64 x 32 sub-tile: 3 ds_read_b128 per 2 MFMAs, which together with
// the LDS-DMA writes keeps the LDS array ~95 % busy (MI355X_MICROARCH.md LDS: a ds_read_b128 is 4 array
// cycles), and the prologue/epilogue of a 9-step tile is as long as its MFMA loop.  This variant:
//   * 4 waves, each a 64-pixel x 64-channel sub-tile of the same 16x16-pixel tile: 1 read per MFMA;
//   * ONE halo buffer (the next channel chunk is fetched after the current one is done) and the same
//     3-slot weight ring: 65 KB of LDS, so TWO workgroups share a CU.
// Measured (64->64 @256x256, B=32): 305 us vs 321 us register-staged / 366 us for the 8-wave form;
// 128->64: 428 vs 494 us.  Ablations: loads alone 80 us, MFMA loop alone 145 us, stores 55 us, statistics
// 23 us -- they add up.  Not a lockstep effect: delaying every second workgroup of the first round (by block
// parity, or by the HW_ID workgroup slot so that the two co-resident ones are half a lifetime apart) changes
// nothing, nor does a third workgroup per CU; the kernel behaves as if bounded by energy (MFMA stream + HBM
// traffic under one power cap; all-zero inputs run measurably faster), which scheduling cannot buy back.  Tried and rejected:
// a persistent 4-wave form with next-tile halo prefetch and a dedicated staging tile (378 us: one wave
// per SIMD serialises LDS-DMA issue, MFMAs and the epilogue), and keeping the 72 KB of weights in
// registers (36-72 B-fragments per wave; hipcc spills around the epilogue and every scratch reload
// drains the DMA queue: 470-610 us).
// (BN = 128 instantiates too -- 64 x 128 per wave, 2-slot ring, 74 KB -- and was the default for N >= 128 until
// the channel-split kernel below beat it by 12 %; CRIMAC_CONV_W4=1 selects it for A/B runs.)
// PP: plane-pair input (p.Cin / p.in_ld count HALVES: twice the channels), 3 MFMAs per fragment pair; TO: output storage
template <int BN, typename T16, int MODE, typename TO = T16, bool PP = false>     // MODE: the epilogue's fused reduction (0 none, 1 statistics, 2 BatchNorm-backward sums)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))      // <= 256 registers: 2 workgroups/CU
void conv3x3_glds_w4_kernel(ConvParams p) {
  constexpr int NW = 4, NT = BN / 16;
  constexpr int NSLOT = BN == 64 ? CRIMAC_W4_NSLOT64 : 2, AHEAD = NSLOT - 1;
  constexpr int B_BYTES = BN * RB;                 // weight slot: 8 / 16 KB
  constexpr int NB = (BN / 8) / NW;                // weight wave-instructions per wave and step: 2 / 4
  constexpr int NH = (HALO_INSTR + NW - 1) / NW;   // 11

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  auto sB = [&](int slot) { return smem + A_BYTES + slot * B_BYTES; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane >> 3, c8 = lane & 7;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  int tile_m = bid;
  const int txi = tile_m % p.tiles_x;
  tile_m /= p.tiles_x;
  const int tyi = tile_m % p.tiles_y;
  const int b = tile_m / p.tiles_y;
  const int y0 = tyi * TR, x0 = txi * TC;
  const int n0 = p.n_first + blockIdx.y * BN;
  const T16* inp = reinterpret_cast<const T16*>(p.in);

  long h_src[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int k = wave + NW * i;
    const int row = 8 * k + sub;
    h_src[i] = -1;
    if (k < HALO_INSTR && row < HALO_ROWS) {
      const int hy = row / HP, hx = row % HP;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      const int u = src_unit<PP>(c8 ^ halo_swz(hx));
      if (y >= 0 && y < p.H && x >= 0 && x < p.W)
        h_src[i] = (((long)b * p.H + y) * p.W + x) * p.in_ld + u * 8;
      else
        *reinterpret_cast<u32x4*>(sA + k * 1024 + lane * 16) = u32x4{0, 0, 0, 0};
    }
  }
  auto issue_halo = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (h_src[i] >= 0) glds16(inp + h_src[i] + kc * BK, sA + (wave + NW * i) * 1024);
  };
  long b_src[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = 8 * (wave + NW * i) + sub;
    b_src[i] = (long)(n0 + row) * p.Cin + ((c8 ^ ((row >> 1) & 7)) * 8);
  }
  const long w_tap = (long)p.N * p.Cin;
  auto issue_b = [&](int kc, int t, int slot) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      glds16(p.w_hi + t * w_tap + b_src[i] + kc * BK, sB(slot) + (wave + NW * i) * 1024);
  };

  // 16x16x32 MFMAs: M tile i = image row 4*wave + i of the pixel tile, N tile j = 16 channels
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](const unsigned char* Bs, int t) {
    const int kx = t % 3;
    const unsigned char* a0 = sA + ((wave * 4 + t / 3) * HP + fr + kx) * RB;
    const int aswz = halo_swz(fr + kx);
    if constexpr (PP) {
      // k-step 0 = hi planes, k-step 1 = lo planes of the same 32 channels: lo*hi + hi*lo + hi*hi (smallest first)
      bf16x8 af[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[ks][i] = *reinterpret_cast<const bf16x8*>(a0 + i * (HP * RB) + (((4 * ks + fq) ^ aswz) << 4));
#pragma unroll
      for (int jh = 0; jh < NT; jh += 4) {
        bf16x8 bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = (jh + j) * 16 + fr;
            bfr[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + row * RB + (((4 * ks + fq) ^ ((row >> 1) & 7)) << 4));
          }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][jh + j] = E16<T16>::mfma16(af[1][i], bfr[0][j], acc[i][jh + j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][jh + j] = E16<T16>::mfma16(af[0][i], bfr[1][j], acc[i][jh + j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][jh + j] = E16<T16>::mfma16(af[0][i], bfr[0][j], acc[i][jh + j]);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int unit = 4 * ks + fq;
      bf16x8 af[4], bfr[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(a0 + i * (HP * RB) + ((unit ^ aswz) << 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = j * 16 + fr;
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * RB + ((unit ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = E16<T16>::mfma16(af[i], bfr[j], acc[i][j]);
    }
    }
  };

  const int kchunks = p.Cin / BK;
  const int nsteps = kchunks * 9;
  issue_halo(0);
  issue_b(0, 0, 0);
  if constexpr (AHEAD >= 2) issue_b(0, 1, 1);
  if constexpr (AHEAD >= 3) issue_b(0, 2, 2);
  wait_vmcnt<(AHEAD - 1) * NB>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int kc = 0, t = 0, slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + AHEAD < nsteps;
    if (more) {
      int t2 = t + AHEAD, kc2 = kc;
      if (t2 >= 9) { t2 -= 9; kc2 += 1; }
      int slot2 = slot + AHEAD;
      if (slot2 >= NSLOT) slot2 -= NSLOT;
      issue_b(kc2, t2, slot2);
    }
#ifndef CRIMAC_EXP_NOCOMPUTE
    compute(sB(slot), t);
#endif
    if (more) wait_vmcnt<(AHEAD - 1) * NB>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (++t == 9) {
      t = 0;
      if (++kc < kchunks) {          // every wave is done with this chunk's halo: fetch the next one
        issue_halo(kc);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (++slot == NSLOT) slot = 0;
  }
#ifdef CRIMAC_EXP_NOEPI
  float sum = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < NT; ++j)
      for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  if (sum == 12345.678f) reinterpret_cast<float*>(p.epi.out)[tid] = sum;
#else
  conv_epilogue<TO, BN, BM, 256, 4, NT, f32x4, MODE, EpiPasses<TO>::value>(acc, p.epi, smem, b, y0, x0, n0, TR, wave, 0);
#endif
}

template <int BN, typename T16, typename TO = T16, bool PP = false>
int launch_w4(ConvParams p, hipStream_t st) {
  p.tiles_y = cdiv(p.H, TR);
  p.tiles_x = cdiv(p.W, TC);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  const size_t lds = (size_t)A_BYTES + (BN == 64 ? CRIMAC_W4_NSLOT64 : 2) * BN * RB;
  static_assert(BM / EpiPasses<TO>::value * (BN * EpiPasses<TO>::kStageBytes + 16) + 2 * BN * 4 <= A_BYTES + 2 * BN * RB,
                "epilogue staging must fit");
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_glds_w4_kernel<BN, T16, 0, TO, PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_glds_w4_kernel<BN, T16, 1, TO, PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (!__is_same(TO, hp_t))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_glds_w4_kernel<BN, T16, 2, TO, PP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid((unsigned)ntiles, p.n_count / BN);
  const int mode = p.epi.stat_sum ? p.epi.stat_mode : 0;
  if constexpr (__is_same(TO, hp_t)) {          // (a plane-pair output is never the `da` of a BatchNorm block: no mode 2)
    CRIMAC_REQUIRE(mode != 2, "conv3x3: plane-pair output with fused BatchNorm-backward sums");
    if (mode == 0) hipLaunchKernelGGL((conv3x3_glds_w4_kernel<BN, T16, 0, TO, PP>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_glds_w4_kernel<BN, T16, 1, TO, PP>), grid, dim3(256), lds, st, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((conv3x3_glds_w4_kernel<BN, T16, 0, TO, PP>), grid, dim3(256), lds, st, p);
    else if (mode == 1) hipLaunchKernelGGL((conv3x3_glds_w4_kernel<BN, T16, 1, TO, PP>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_glds_w4_kernel<BN, T16, 2, TO, PP>), grid, dim3(256), lds, st, p);
  }
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}


// ---- hand-placed LDS-read / MFMA pipeline of the channel-split kernel -----------------------------------
// A chunk (9 taps x 2 k-steps of 32) is 72 groups of 4 fragment reads + 8 MFMAs; the reads of group h+1 are
// issued before the MFMAs of group h (counted lgkmcnt: LDS reads return in order), across k-steps and taps
// alike -- there is no barrier inside a chunk.  All addresses are one of 6 registers (kx, ks) + immediate.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128_asm(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <bool LAST, typename F>
__device__ __forceinline__ void wch_release(F& f, int set) {
  if constexpr (LAST)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[set][0]), "+v"(f.a[set][1]), "+v"(f.a[set][2]), "+v"(f.a[set][3])
                 :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.a[set][0]), "+v"(f.a[set][1]), "+v"(f.a[set][2]), "+v"(f.a[set][3])
                 :: "memory");
}
__device__ __forceinline__ void wch_land_b(bf16x8 (&bf)[4]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]) :: "memory");
}

// ---------------------------------------------------------------------------------------------------
// Channel-split kernel for N % 128 == 0 ("wch"): each of the 4 waves owns 32 OUTPUT CHANNELS of the whole
// 16x16-pixel tile.  Its weight fragments are then private to the wave and come straight from global memory
// into registers (4 x 16 B per lane and tap, prefetched one tap ahead) -- no weight tile in LDS, no weight DMA,
// no weight ds_reads -- and the only LDS content is the halo, which is constant for a whole 64-channel chunk:
// NO barrier per tap, the waves run free for 576 MFMAs and meet only where the halo is exchanged.
// MFMA shape: v_mfma_f32_16x16x32_bf16, 16 x 2 tiles per wave -- an M tile is one image row of the pixel tile,
// every A fragment feeds two MFMAs (the two 16-channel halves).  The 32x32x16 form of this kernel (8 x 1 tiles,
// same reads, same registers, same cycle count) was 7-11 % SLOWER on every layer (1.40 vs 1.58 PFLOP/s on
// 1024->512 @32x32): under a dense MFMA stream the chip holds a higher clock with the 16x16x32 shape
// (MI355X_MICROARCH.md, clocks under load (7)).
// Measured motivation for the structure (W4 kernel, 1024->512 @32x32): MFMA loop alone 1.7 PFLOP/s, with the
// weight/halo DMAs sharing the LDS 1.2 PFLOP/s; here the LDS sees 1 ds_read_b128 per 2 MFMAs and 4.6 KB of DMA
// per step.  Tried on top and rejected: 32-channel chunks with a double-buffered halo (next halo requested at
// the start of a chunk behind the weight loads of two taps, counted vmcnt, one barrier per chunk): 2-4 % SLOWER
// -- the halo exchange bubble is already covered by the second workgroup of the CU; a 2x2-wave (pixels x
// channels) form for N = 64: 319 / 458 us at two workgroups per CU, 297 / 425 us at three (168 registers, 3
// spilled) vs 297 / 427 us for the pixel-split kernel -- no gain, not used.
struct WchFrags {
  bf16x8 a[2][4];
  bf16x8 b[2][4];        // [tap parity][ks2 * 2 + nb]
};
// NQ: groups of 4 image rows per wave (4: the wave covers all 16 rows of the tile; 2: the 2 x 2 form for 64-channel
// tiles, waves = 2 pixel halves x 2 channel halves, the half's row offset rides in the address registers)
// ILV (tall form): the wave's second half of row groups lies 2 NQ image rows further down -- a wave owns rows
// [8 wp, 8 wp + 8) and [16 + 8 wp, 16 + 8 wp + 8) of the 32-row tile, so that each 16-row slice of the epilogue holds half of
// EVERY wave's accumulators (with 16 contiguous rows per wave one wave pair stages a slice while the other idles)
template <int H, int NQ = 4, bool ILV = false>
__device__ __forceinline__ void wch_issue(const unsigned (&av)[3][2], WchFrags& f) {
  constexpr int t = H / (2 * NQ), ks2 = (H / NQ) % 2, q = H % NQ;
  constexpr int base = (t / 3 + 4 * q + (ILV && q >= NQ / 2 ? 2 * NQ : 0)) * (HP * RB);
  f.a[H & 1][0] = lds_read128_asm<base + 0 * (HP * RB)>(av[t % 3][ks2]);
  f.a[H & 1][1] = lds_read128_asm<base + 1 * (HP * RB)>(av[t % 3][ks2]);
  f.a[H & 1][2] = lds_read128_asm<base + 2 * (HP * RB)>(av[t % 3][ks2]);
  f.a[H & 1][3] = lds_read128_asm<base + 3 * (HP * RB)>(av[t % 3][ks2]);
}
// WF (fragment-major weight plane, common.h wfrag_index): the four fragments of a (tap, 32-channel block, 64-deep chunk) are
// four CONSECUTIVE kilobytes, lane l at byte 16 l of each -- every load instruction reads whole 128-byte lines (the row-major
// plane [tap][N][Cin] gives a wave instruction 16 rows x 64 bytes: sixteen half lines).  One address register, four
// immediates.  Measured as an ablation first (round 5, r5_08: -2.5 % over the conv launches of a step, -12 % on the
// 1024-channel bottleneck layers, whose workgroups stream 2.4 MB of weights each).
template <bool WF = false>
__device__ __forceinline__ void wch_load_b(const unsigned short* s0, const unsigned short* s1, bf16x8 (&bf)[4]) {
#ifdef CRIMAC_EXP_WCH_NOW        // (ablation build: no weight stream -- the MFMAs run on whatever the registers hold)
  asm volatile("" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]) : "v"(s0), "v"(s1));
  return;
#endif
#ifdef CRIMAC_EXP_WCH_HALFW       // (ablation build, results garbage: HALF the weight stream -- two of the four fragments per tap)
  if constexpr (WF) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %4, off offset:1024"
        : "=&v"(bf[0]), "=&v"(bf[1]), "+v"(bf[2]), "+v"(bf[3])
        : "v"(s0)
        : "memory");
    return;
  }
#endif
  if constexpr (WF) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %4, off offset:1024\n\t"
        "global_load_dwordx4 %2, %4, off offset:2048\n\t"
        "global_load_dwordx4 %3, %4, off offset:3072"
        : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2]), "=&v"(bf[3])
        : "v"(s0)
        : "memory");
  } else {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %5, off\n\t"
        "global_load_dwordx4 %2, %4, off offset:64\n\t"
        "global_load_dwordx4 %3, %5, off offset:64"
        : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2]), "=&v"(bf[3])
        : "v"(s0), "v"(s1)
        : "memory");
  }
}
// PP (plane pairs): k-step 0 of a chunk is the hi plane of 32 channels, k-step 1 their lo plane, in both operands: the
// hi fragments of A meet both weight planes (hi*lo, then hi*hi), the lo fragments the hi weights only -- 3 MFMAs per
// fragment pair, 16 + 8 per pair of groups, on the same reads and weight loads as the 16-bit kernel's 8 + 8.
template <typename T16, bool PP, int H, int NQ = 4, bool ILV = false, bool WF = false, typename ACC>
__device__ __forceinline__ void wch_step(const unsigned (&av)[3][2], const unsigned short* wtap, long w_tap, long w_nb,
                                           const unsigned short* wnext_chunk, WchFrags& f, ACC& acc) {
  constexpr int t = H / (2 * NQ), ks2 = (H / NQ) % 2, q = H % NQ, NH = 18 * NQ;
  if constexpr (H % (2 * NQ) == 0) {
    if constexpr (t == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) f.b[0][k] = f.b[1][k];
    } else {
      wch_land_b(f.b[t & 1]);
    }
    if constexpr (t < 8) {
      const unsigned short* s = wtap + (t + 1) * w_tap;
      wch_load_b<WF>(s, s + w_nb, f.b[(t + 1) & 1]);
    } else {
      const unsigned short* s = wnext_chunk ? wnext_chunk : wtap;
      wch_load_b<WF>(s, s + w_nb, f.b[1]);
    }
  }
#ifdef CRIMAC_EXP_WCH_FEWA        // (ablation build, results garbage: 3 of 8 groups of LDS fragment reads -- what a form that uses
  //                                  every halo-row fragment for the three taps of its column would read)
  if constexpr (H + 1 < NH && (H + 1) % 8 < 3) {
    wch_issue<H + 1, NQ, ILV>(av, f);
    wch_release<false>(f, H & 1);
  } else {
    wch_release<true>(f, H & 1);
  }
#else
  if constexpr (H + 1 < NH) {
    wch_issue<H + 1, NQ, ILV>(av, f);
    wch_release<false>(f, H & 1);
  } else {
    wch_release<true>(f, H & 1);
  }
#endif
  if constexpr (PP && ks2 == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        acc[4 * q + j][nb] = E16<T16>::mfma16(f.a[H & 1][j], f.b[t & 1][2 + nb], acc[4 * q + j][nb]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
      acc[4 * q + j][nb] = E16<T16>::mfma16(f.a[H & 1][j], f.b[t & 1][(PP ? 0 : ks2 * 2) + nb],
                                                                    acc[4 * q + j][nb]);
  if constexpr (H + 1 < NH) wch_step<T16, PP, H + 1, NQ, ILV, WF>(av, wtap, w_tap, w_nb, wnext_chunk, f, acc);
}

// FORM 3 ("rows"): 64 output channels of a 32 x 16-pixel tile, waves = 4 channel QUARTERS (16 channels each) of all 32
// image rows.  Same accumulators per wave as the other forms (32 M tiles x 1 channel tile), HALF the weight stream per
// MFMA -- a (tap, k-step) fragment of 16 channels feeds the 32 MFMAs of the wave's 512 pixels, not 16 -- which is what
// round 5's ablations priced at 13-15 % of the channel-split launches (profiles/r05_wch_weight_stream_half_same.txt); the
// halo fragments a 16-channel wave needs per MFMA would double, so the loop runs over HALO rows instead of taps: a
// (column tap kx, k-step) PHASE walks the 34 halo rows once, and the fragment of halo row r meets the weights of the three
// row taps ky at output rows r - ky: 34 ds_read_b128 per 96 MFMAs (the tap loop: 48), which the LDS carries easily
// (profiles/r05_wch_lds_reads_ablation.txt: 3/8 of the reads buy 2.5 %, the reads are not what the kernel waits for).
// Weights: fragment-major planes only; phase (kx, ks) loads three fragments (ky = 0..2) one phase (96 MFMAs) ahead.
struct RowFrags {
  bf16x8 a[4];           // halo-row fragments, read RD_AHEAD rows ahead
  bf16x8 b[2][3];        // [phase parity][ky]
};
constexpr int kRowSteps = 6 * 34, kRowAhead = 3;
template <int S>
__device__ __forceinline__ void wrow_issue(const unsigned (&av)[3][2], const unsigned (&avh)[3][2], RowFrags& f) {
  constexpr int ph = S / 34, r = S % 34, kx = ph >> 1, ks = ph & 1;
  if constexpr (r < 17) f.a[S & 3] = lds_read128_asm<r * (HP * RB)>(av[kx][ks]);
  else f.a[S & 3] = lds_read128_asm<(r - 17) * (HP * RB)>(avh[kx][ks]);
}
template <int N>
__device__ __forceinline__ void wrow_release(bf16x8& a) {
  static_assert(N >= 0 && N <= 3, "reads in flight behind the one awaited");
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a) :: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a) :: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a) :: "memory");
}
__device__ __forceinline__ void wrow_load_b(const unsigned short* s0, const unsigned short* s1, const unsigned short* s2,
                                            bf16x8 (&bf)[3]) {
  asm volatile(
      "global_load_dwordx4 %0, %3, off\n\t"
      "global_load_dwordx4 %1, %4, off\n\t"
      "global_load_dwordx4 %2, %5, off"
      : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2])
      : "v"(s0), "v"(s1), "v"(s2)
      : "memory");
}
__device__ __forceinline__ void wrow_land_b(bf16x8 (&bf)[3]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]) :: "memory");
}
// wch: this wave's fragment (lane's 16 bytes) of tap 0, k-step 0 of the current 64-channel chunk; w_tap: halves per tap
template <typename T16, int S, typename ACC>
__device__ __forceinline__ void wrow_step(const unsigned (&av)[3][2], const unsigned (&avh)[3][2], const unsigned short* wch,
                                          long w_tap, const unsigned short* wnext_chunk, RowFrags& f, ACC& acc) {
  constexpr int ph = S / 34, r = S % 34;
  if constexpr (r == 0) {
    wrow_land_b(f.b[ph & 1]);
    if constexpr (ph < 5) {
      constexpr int kx = (ph + 1) >> 1, ks = (ph + 1) & 1;
      const unsigned short* s = wch + kx * w_tap + ks * 1024;
      wrow_load_b(s, s + 3 * w_tap, s + 6 * w_tap, f.b[(ph + 1) & 1]);
    } else {
      const unsigned short* s = wnext_chunk ? wnext_chunk : wch;
      wrow_load_b(s, s + 3 * w_tap, s + 6 * w_tap, f.b[0]);
    }
  }
  if constexpr (S + kRowAhead < kRowSteps) wrow_issue<S + kRowAhead>(av, avh, f);
  wrow_release<(kRowSteps - 1 - S < kRowAhead ? kRowSteps - 1 - S : kRowAhead)>(f.a[S & 3]);
  if constexpr (r < 32) acc[r < 32 ? r : 0][0] = E16<T16>::mfma16(f.a[S & 3], f.b[ph & 1][0], acc[r < 32 ? r : 0][0]);
  if constexpr (r >= 1 && r < 33) acc[r >= 1 && r < 33 ? r - 1 : 0][0] = E16<T16>::mfma16(f.a[S & 3], f.b[ph & 1][1], acc[r >= 1 && r < 33 ? r - 1 : 0][0]);
  if constexpr (r >= 2) acc[r >= 2 ? r - 2 : 0][0] = E16<T16>::mfma16(f.a[S & 3], f.b[ph & 1][2], acc[r >= 2 ? r - 2 : 0][0]);
  if constexpr (S + 1 < kRowSteps) wrow_step<T16, S + 1>(av, avh, wch, w_tap, wnext_chunk, f, acc);
}

// FORM 1 ("S22"): the 2 x 2 form for 64-channel tiles: waves = 2 pixel halves (8 image rows each) x 2 channel halves (32
// channels each); the wave's weight fragments still come straight from global memory (each half is fetched by two waves).
// FORM 2 ("tall"): 64 output channels of a 32 x 16-pixel tile: waves = 2 pixel halves of SIXTEEN image rows x 2 channel
// halves -- every wave is exactly the wave of the 128-channel form (16 M tiles x 32 channels, the same reads, weight
// loads and MFMAs per tap), so a weight fragment feeds as many MFMAs as there; what differs is the halo (34 x 18 pixels,
// 76.5 KB per chunk: 1.9x the bytes per MFMA, which is what 64 output channels cost) -- two workgroups per CU.
// Why: in FORM 1 a wave's tap is 48 MFMAs (plane pairs) between two weight-fragment waits and a workgroup's prologue,
// halo waits and epilogue (28 k cycles) stand against 13.8 k cycles of MFMA work per SIMD: MFMA busy 0.45 (PMC) where the
// 128-channel form reaches 0.77.
template <typename T16, int MODE, typename TO = T16, bool PP = false, int FORM = 0, bool WF = false>      // MODE: the epilogue's fused reduction (0 none, 1 BatchNorm statistics, 2 BatchNorm-backward sums); WF: fragment-major weight plane
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FORM == 1 ? 3 : 2, FORM == 1 ? 3 : 2))) void conv3x3_wch_kernel(ConvParams p) {
  constexpr bool S22 = FORM != 0;                  // 64-channel tiles (forms 1, 2: waves = pixel halves x channel halves; 3: channel quarters)
  constexpr bool ROWS = FORM == 3;
  static_assert(!ROWS || (WF && !PP), "the rows form reads fragment-major 16-bit planes");
  constexpr int BN = S22 ? 64 : 128, NW = 4;
  constexpr int NQ = FORM == 1 ? 2 : 4, WR = ROWS ? 32 : 4 * NQ;       // image rows (16-pixel M tiles) per wave
  constexpr int WN = ROWS ? 1 : 2;                         // 16-channel tiles per wave
  constexpr int TRK = FORM >= 2 ? 32 : TR;                 // image rows of the workgroup's tile
  constexpr bool ILV = FORM == 2;                          // the wave's rows: two blocks of WR / 2 (wch_issue)
  constexpr int HALO_ROWS_K = (TRK + 2) * HP;              // 324 (612)
  constexpr int HALO_INSTR_K = (HALO_ROWS_K + 7) / 8;      // 41 (77) wave-instructions of 8 rows
  constexpr int NH = (HALO_INSTR_K + NW - 1) / NW;         // 11 (20)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CRIMAC_DIAG_PHASES
  unsigned long long wph[4] = {0, 0, 0, 0}, wph_t, wrt0;
  { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wph_t), "=s"(wrt0) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define CRIMAC_CPH(k) { unsigned long long tn; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    wph[k] += tn - wph_t; wph_t = tn; }
#else
#define CRIMAC_CPH(k)
#endif
  unsigned char* sA = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane >> 3, c8 = lane & 7;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  int tile_m = bid;
  const int txi = tile_m % p.tiles_x;
  tile_m /= p.tiles_x;
  const int tyi = tile_m % p.tiles_y;
  const int b = tile_m / p.tiles_y;
  const int y0 = tyi * TRK, x0 = txi * TC;
  const int n0 = p.n_first + blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.in), 0, (int)((((long)p.B * p.H * p.W - 1) * p.in_ld + p.Cin) * 2), 0x00020000);
  // per-lane source offset of halo DMA instruction i (8 halo rows each): a table in registers; the tall form (20
  // instructions per wave, and the accumulators of the 128-channel form) recomputes it at every chunk instead -- ~12
  // integer operations per instruction next to 864 MFMAs, against 20 registers that the epilogue would spill
  auto halo_off = [&](int i) -> unsigned {
    const int k = wave + NW * i;
    int row = 8 * k + sub;
    if constexpr (FORM >= 2) asm volatile("" : "+v"(row));          // (recomputed where it is used, not hoisted back into a table)
    const int hy = (row * 3641) >> 16, hx = row - hy * HP;          // row / 18 (exact for row < 4096)
    const unsigned y = (unsigned)(y0 + hy - 1), x = (unsigned)(x0 + hx - 1);
    const bool ok = k < HALO_INSTR_K && row < HALO_ROWS_K && y < (unsigned)p.H && x < (unsigned)p.W;
    // (the launcher guarantees 32-bit byte offsets: `small`)
    return ok ? (unsigned)((((b * p.H + (int)y) * p.W + (int)x) * (int)p.in_ld + src_unit<PP>(c8 ^ halo_swz(hx)) * 8) * 2)
              : 0x80000000u;
  };
  constexpr bool HTAB = FORM < 2;
  unsigned h_off[HTAB ? NH : 1];
  if constexpr (HTAB) {
#pragma unroll
    for (int i = 0; i < NH; ++i) h_off[i] = halo_off(i);
  }
  auto issue_halo = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (wave + NW * i < HALO_INSTR_K) {
        unsigned off;
        if constexpr (HTAB) off = h_off[i]; else off = halo_off(i);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(sA + (wave + NW * i) * 1024), 16, (int)off,
            kc * BK * 2, 0, CRIMAC_WCH_HALO_AUX);
      }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int wc = S22 && !ROWS ? (wave & 1) : wave, wp = S22 && !ROWS ? (wave >> 1) : 0;
  // weight fragments of this lane: rows n0 + 32*wc + nb*16 + fr of tap t, k = kc*64 + ks2*32 + fq*8 .. +8
  // (WF: block (n0 / 32 + wc) of the tap's N / 32 blocks, Cin / 64 chunks of 2048 halves each, lane l at 8 l)
#ifdef CRIMAC_EXP_WCH_SAMEW       // (ablation build, results garbage: EVERY wave streams channel block 0 -- the same bytes per wave,
  //                                  all of them L1 / L2 hits: separates the L2 -> L1 traffic from the delivery into the CU)
  const unsigned short* wrow = p.w_hi + lane * 8;
#else
  const unsigned short* wrow = ROWS ? p.w_hi + ((long)((n0 >> 5) + (wave >> 1)) * (p.Cin >> 6)) * 2048 + (wave & 1) * 512 + lane * 8
                               : WF ? p.w_hi + ((long)((n0 >> 5) + wc) * (p.Cin >> 6)) * 2048 + lane * 8
                                    : p.w_hi + (long)(n0 + 32 * wc + fr) * p.Cin + fq * 8;
#endif
  const long w_tap = (long)p.N * p.Cin, w_nb = 16L * p.Cin;
  constexpr int W_CHUNK = WF ? 2048 : BK;             // halves between two 64-channel chunks of a tap
  // A fragment: M tile i = image row i of the tile, lane's pixel column fr -> halo row (i + ky) * HP + fr + kx
  unsigned av[3][2];
  {
    const unsigned a_lds = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(sA));
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2)
        av[kx][ks2] = a_lds + (wp * (ILV ? WR / 2 : WR) * HP + fr + kx) * RB + (((4 * ks2 + fq) ^ halo_swz(fr + kx)) << 4);
  }

  f32x4 acc[WR][WN];
#pragma unroll
  for (int i = 0; i < WR; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  const int kchunks = p.Cin / BK;
#ifndef CRIMAC_WCH_NO_BIAS_PRE
  // this lane's bias values, requested here and used by the epilogue (conv_epilogue.h: bias_pre)
  float bias_pre[WN];
#pragma unroll
  for (int nb = 0; nb < WN; ++nb) bias_pre[nb] = p.epi.bias ? p.epi.bias[n0 + wc * (16 * WN) + nb * 16 + (lane & 15)] : 0.f;
#endif
  unsigned avh[3][2];                    // (rows form) halo rows 17 ..: a ds_read's immediate offset ends at 64 KB
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) avh[kx][ks2] = av[kx][ks2] + 17 * (HP * RB);
  WchFrags f;
  RowFrags rf;
  issue_halo(0);                         // (in flight together with the first weight fragments: one latency, not two)
  if constexpr (ROWS) {
    wrow_load_b(wrow, wrow + 3 * w_tap, wrow + 6 * w_tap, rf.b[0]);
    wrow_land_b(rf.b[0]);
  } else {
    wch_load_b<WF>(wrow, wrow + w_nb, f.b[1]);
    wch_land_b(f.b[1]);                  // vmcnt(0): the fragments and the first halo chunk
  }
  CRIMAC_DIAG_STAMP(dg_t0, dg_r0)
  CRIMAC_CPH(0)
  for (int kc = 0; kc < kchunks; ++kc) {
    if (kc > 0) {
      issue_halo(kc);
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CRIMAC_CPH(1)
    const unsigned short* wtap = wrow + (long)kc * W_CHUNK;
    const unsigned short* wnext = kc + 1 < kchunks ? wtap + W_CHUNK : nullptr;
    if constexpr (ROWS) {
      wrow_issue<0>(av, avh, rf);
      wrow_issue<1>(av, avh, rf);
      wrow_issue<2>(av, avh, rf);
      wrow_step<T16, 0>(av, avh, wtap, w_tap, wnext, rf, acc);
      wrow_land_b(rf.b[0]);
    } else {
      wch_issue<0, NQ, ILV>(av, f);
      wch_step<T16, PP, 0, NQ, ILV, WF>(av, wtap, w_tap, w_nb, wnext, f, acc);
      wch_land_b(f.b[1]);
    }
    __builtin_amdgcn_s_barrier();
    CRIMAC_CPH(2)
  }
  CRIMAC_DIAG_STAMP(dg_t1, dg_r1)
#ifndef CRIMAC_DIAG_PHASES
  CRIMAC_DIAG_STORE(crimac_diag_clock_conv, dg_t0, dg_r0, dg_t1, dg_r1)
#endif
  // (tall form, 16-bit output: two slices of 256 rows like the other forms' whole tile -- 16 fully unrolled store rounds
  // spill ~60 registers in the statistics mode)
  constexpr int EPASS = EpiPasses<TO>::value * (FORM >= 2 && sizeof(TO) == 2 ? 2 : 1);
#ifndef CRIMAC_WCH_NO_BIAS_PRE
  conv_epilogue<TO, BN, TRK * TC, 256, WR, WN, f32x4, MODE, EPASS, ILV>(acc, p.epi, smem, b, y0, x0, n0, TRK, wp, wc, bias_pre);
#else
  conv_epilogue<TO, BN, TRK * TC, 256, WR, WN, f32x4, MODE, EPASS, ILV>(acc, p.epi, smem, b, y0, x0, n0, TRK, wp, wc);
#endif
#ifdef CRIMAC_DIAG_PHASES
  // cycles of wave 0: prologue | waiting for the halo chunks | MFMA steps | epilogue (stores issued)
  CRIMAC_CPH(3)
  const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) % 1024;
  if (tid == 0)
    for (int k = 0; k < 4; ++k) crimac_diag_clock_conv_buf[slot * 4 + k] = wph[k];
  // wall clock (100 MHz) at the workgroup's entry and exit, and where it ran: [4096 + 2 slot] = entry, [+ 1] = exit << 16 | XCC_ID << 8 | CU
  {
    unsigned long long wrt1;
    unsigned hwid, xcc;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(wrt1), "=s"(hwid), "=s"(xcc) :: "memory");
    if (tid == 0) {
      crimac_diag_clock_conv_buf[4096 + 2 * slot] = wrt0;
      crimac_diag_clock_conv_buf[4096 + 2 * slot + 1] = (wrt1 << 16) | ((unsigned long long)(xcc & 15) << 8) | ((hwid >> 8) & 15) | (((hwid >> 13) & 7) << 4);
    }
  }
#endif
}

template <typename T16, typename TO = T16, bool PP = false, int S22 = 0, bool WF = false>      // S22: the kernel's FORM
int launch_wch(ConvParams p, hipStream_t st) {
  constexpr int BN = S22 ? 64 : 128;
  constexpr int TRK = S22 >= 2 ? 32 : TR;
  constexpr int HALO_BYTES = ((TRK + 2) * HP + 7) / 8 * 1024;      // one halo buffer, padded to whole DMA instructions
  p.tiles_y = cdiv(p.H, TRK);
  p.tiles_x = cdiv(p.W, TC);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  constexpr int EPASS = EpiPasses<TO>::value * (S22 >= 2 && sizeof(TO) == 2 ? 2 : 1);      // (as in the kernel)
  constexpr size_t stage = (size_t)(TRK * TC / EPASS) * (BN * EpiPasses<TO>::kStageBytes + 16) + 2 * BN * 4;
  const size_t lds = stage > (size_t)HALO_BYTES ? stage : (size_t)HALO_BYTES;
  static_assert(stage <= 80 * 1024 && HALO_BYTES <= 80 * 1024, "two workgroups per CU");
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wch_kernel<T16, 0, TO, PP, S22, WF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wch_kernel<T16, 1, TO, PP, S22, WF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (!__is_same(TO, hp_t))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wch_kernel<T16, 2, TO, PP, S22, WF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid((unsigned)ntiles, p.n_count / BN);
  const int mode = p.epi.stat_sum ? p.epi.stat_mode : 0;
  if constexpr (__is_same(TO, hp_t)) {
    CRIMAC_REQUIRE(mode != 2, "conv3x3: plane-pair output with fused BatchNorm-backward sums");
    if (mode == 0) hipLaunchKernelGGL((conv3x3_wch_kernel<T16, 0, TO, PP, S22, WF>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_wch_kernel<T16, 1, TO, PP, S22, WF>), grid, dim3(256), lds, st, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((conv3x3_wch_kernel<T16, 0, TO, PP, S22, WF>), grid, dim3(256), lds, st, p);
    else if (mode == 1) hipLaunchKernelGGL((conv3x3_wch_kernel<T16, 1, TO, PP, S22, WF>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_wch_kernel<T16, 2, TO, PP, S22, WF>), grid, dim3(256), lds, st, p);
  }
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// ---------------------------------------------------------------------------------------------------
// First layer (unet.py:77 with in_channels = 4, padded to 16): K = 9 x 16 = 144 -- 80 MFMAs per wave and tile,
// the kernel is its output stream (32 KB + BatchNorm statistics per tile against 10 KB of input).  The generic
// register-staged kernel spent 9 staged K steps with two barriers each on it (181 us at B = 32).  Measured on
// the way here (one tile per workgroup, shared epilogue): 145 us without / 180 us with statistics, of which
// the halo + weight loads 50 us (30 us alone: their latency is not hidden by anything) and the per-tile
// statistics atomics 35 us.  Hence a PERSISTENT kernel:
//   * all weights (18 KB, L2) go to LDS ONCE per workgroup; 5 k-steps of v_mfma_f32_16x16x32_bf16 pair two taps
//     per step (k = 16 channels of tap 2s | 16 channels of tap 2s+1; the tenth half-step has zero weights);
//   * the next tile's halo (18x18 pixels x 32 B) is fetched into registers before the current tile's MFMAs and
//     output, so its latency hides behind them;
//   * wave w owns image rows 4w .. 4w+3 x all 64 channels and writes them through a private LDS slab, one
//     16-pixel row (2 KB contiguous in HBM) at a time -- no workgroup barrier in the epilogue;
//   * the statistics stay in registers across tiles and are flushed once per workgroup.
// PP (plane pairs, CRIMAC_PREC_H3P): the SAME 16-channel fp16 convolution on sixteen PSEUDO-channels per pixel and tap,
//   input  [x_hi(4) | x_lo(4) | x_hi(4) | 0(4)]   (the four real channels' hi / lo planes, assembled while the halo is staged
//                                                  from the [8 hi][8 lo] groups in memory),
//   weight [w_hi(4) | w_hi(4) | w_lo(4) | 0(4)]   (assembled when the weights go to LDS, from the interleaved [16 hi | 16 lo] rows),
// whose sum over the sixteen "channels" is x_hi w_hi + x_lo w_hi + x_hi w_lo -- the three products of the plane-pair kernels
// in the k-steps the 16-bit first layer already runs (the generic register-staged kernel took 232 us forward, 291 us in
// inference).  TO: output storage (PP: float = the training forward's y, hp_t = plane pairs for inference).
template <typename T16, typename TO = T16, bool PP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(TO) == 2 ? 4 : 3, sizeof(TO) == 2 ? 4 : 3)))      // <= 128 (168) registers: 4 (3) workgroups/CU
void conv3x3_c16_kernel(ConvParams p, int ntiles) {
  constexpr bool HPO = __is_same(TO, hp_t);
  using TS = typename std::conditional<HPO, float, TO>::type;      // element type of the output slabs
  constexpr int BN = 64, CB = 32;                 // bytes per halo pixel
  constexpr int NHU = (HALO_ROWS * 2 + 255) / 256;   // halo 16-byte units per thread: 3
  constexpr int SLAB_PITCH = BN * (int)sizeof(TS) + 16, SLAB = 16 * SLAB_PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* slab = smem + HALO_ROWS * CB + wave * SLAB;
  float* sstat = reinterpret_cast<float*>(smem + HALO_ROWS * CB + 4 * SLAB);      // [2][64]
  unsigned char* wlds = smem + HALO_ROWS * CB + 4 * SLAB + 2 * BN * 4 + BN * 4;      // (statistics [2][64], bias [64])
  const EpiParams& e = p.epi;
  const T16* inp = reinterpret_cast<const T16*>(p.in);
  TO* outp = reinterpret_cast<TO*>(e.out);
  const bool stats = e.stat_sum != nullptr && e.stat_mode == 1;
  if (tid < 2 * BN) sstat[tid] = 0.f;
  // SW (16-bit outputs): the MFMAs take the WEIGHT fragment as the A operand, so that a lane holds four consecutive CHANNELS
  // of one pixel -- the tile is staged with one packed 8-byte LDS write per accumulator tile instead of four conversions and
  // four 2-byte writes (this kernel is its output stream: ~12 vector instructions per element of staging were most of a
  // tile's cycles at four waves per SIMD).  The bias starts the accumulators (from LDS), the statistics are taken where the
  // slab is read back (this lane's fixed 8-channel chunk).
  constexpr bool SW = sizeof(TS) == 2 && !PP;
  float* sbias = sstat + 2 * BN + 0;              // [64] (SW), behind the statistics; the weights start 256 B further
  if constexpr (SW) {
    if (tid < BN) sbias[tid] = e.bias ? e.bias[tid] : 0.f;
  }

  // weights [tap][n][16] (18 KB) -> LDS once per workgroup; fragment (s, j) of a lane = row j*16 + fr of tap
  // 2s + (fq >> 1), channels (fq & 1)*8 .. +8 (same 32-byte pitch as the halo: conflict-free ds_read_b128)
  const int fr = lane & 15, fq = lane >> 4;
  for (int u = tid; u < 9 * BN * 2; u += 256) {
    if constexpr (PP) {
      // row (tap, n) in memory: [16 hi | 16 lo] halves; unit 0 of its LDS row = [hi 0-3 | hi 0-3], unit 1 = [lo 0-3 | 0]
      const unsigned short* wr = p.w_hi + (long)(u >> 1) * 32;
      const u32x2 hi = *reinterpret_cast<const u32x2*>(wr), lo = *reinterpret_cast<const u32x2*>(wr + 16);
      *reinterpret_cast<u32x4*>(wlds + u * 16) = (u & 1) ? u32x4{lo[0], lo[1], 0u, 0u} : u32x4{hi[0], hi[1], hi[0], hi[1]};
    } else {
      *reinterpret_cast<u32x4*>(wlds + u * 16) = *reinterpret_cast<const u32x4*>(p.w_hi + (long)u * 8);
    }
  }
  int b_off[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    int t = 2 * s + (fq >> 1);
    if (t > 8) t = 8;
    b_off[s] = (t * BN + fr) * CB + (fq & 1) * 16;
  }
  const bool tap9 = (fq >> 1) == 1;             // this lane's half of k-step 4 is the non-existent tenth tap
  float bv[4], cs1[4], cs2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = (!SW && e.bias) ? e.bias[j * 16 + fr] : 0.f;
    cs1[j] = 0.f;
    cs2[j] = 0.f;
  }
  float s1[8], s2[8];                              // (SW) statistics of channel chunk lane & 7
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  // A fragment addresses per k-step (tap 2s + (fq >> 1), clamped where the weights are zero)
  int a_off[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    int t = 2 * s + (fq >> 1);
    if (t > 8) t = 8;
    a_off[s] = ((wave * 4 + t / 3) * HP + fr + t % 3) * CB + (fq & 1) * 16;
  }

  auto tile_geo = [&](int tile, int& b, int& y0, int& x0) {
    const int txi = tile % p.tiles_x;
    const int tt = tile / p.tiles_x;
    b = tt / p.tiles_y;
    y0 = (tt % p.tiles_y) * TR;
    x0 = txi * TC;
  };
  u32x4 hreg[NHU];
  auto fetch_halo = [&](int tile) {
    int b, y0, x0;
    tile_geo(tile, b, y0, x0);
#pragma unroll
    for (int i = 0; i < NHU; ++i) {
      const int u = tid + i * 256;
      const int row = u >> 1, half = u & 1;
      const int hy = row / HP, hx = row - hy * HP;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      hreg[i] = u32x4{0, 0, 0, 0};
      if (u < HALO_ROWS * 2 && y >= 0 && y < p.H && x >= 0 && x < p.W) {
        if constexpr (PP) {
          // pixel in memory (in_ld halves): [8 hi | 8 lo] of channels 0-7 first; unit 0 = [hi 0-3 | lo 0-3], unit 1 = [hi 0-3 | 0]
          const T16* px = inp + (((long)b * p.H + y) * p.W + x) * p.in_ld;
          const u32x2 hi = *reinterpret_cast<const u32x2*>(px);
          if (half == 0) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(px + 8);
            hreg[i] = u32x4{hi[0], hi[1], lo[0], lo[1]};
          } else {
            hreg[i] = u32x4{hi[0], hi[1], 0u, 0u};
          }
        } else {
          hreg[i] = *reinterpret_cast<const u32x4*>(inp + (((long)b * p.H + y) * p.W + x) * p.in_ld + half * 8);
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch_halo(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    int b, y0, x0;
    tile_geo(tile, b, y0, x0);
    __syncthreads();                            // every wave is done reading the previous halo
#pragma unroll
    for (int i = 0; i < NHU; ++i) {
      const int u = tid + i * 256;
      if (u < HALO_ROWS * 2) *reinterpret_cast<u32x4*>(halo + u * 16) = hreg[i];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch_halo(tile + gridDim.x);

    // two image rows (M tiles) at a time: 32 accumulator registers, the weight fragments are re-read from LDS
#pragma unroll 1
    for (int ih = 0; ih < 2; ++ih) {
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (SW) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + j * 16 + fq * 4);      // channels j*16 + fq*4 + r
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        }
      const unsigned char* hrow = halo + ih * (2 * HP * CB);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        bf16x8 af[2], bw[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(hrow + a_off[s] + i * (HP * CB));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bw[j] = *reinterpret_cast<const bf16x8*>(wlds + b_off[s] + j * (16 * CB));
          if (s == 4 && tap9) bw[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = SW ? E16<T16>::mfma16(bw[j], af[i], acc[i][j]) : E16<T16>::mfma16(af[i], bw[j], acc[i][j]);
      }
      // output: image row y0 + 4*wave + 2*ih + i = M tile i; accumulator element r of N tile j is pixel
      // (lane >> 4)*4 + r, channel j*16 + fr
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = y0 + wave * 4 + ih * 2 + i;
        if constexpr (SW) {
          // accumulator element r of N tile j: pixel fr, channel j*16 + fq*4 + r (bias already inside: this kernel has no
          // other form its results would have to equal bit for bit, and reading the bias here costs 10 registers: 74 -> 90 us)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            TS q4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[i][j][r];
              if (e.relu) v = fmaxf(v, 0.f);
              q4[r] = (TS)v;
            }
            *reinterpret_cast<u32x2*>(slab + fr * SLAB_PITCH + (j * 16 + fq * 4) * 2) = *reinterpret_cast<const u32x2*>(q4);
          }
        } else
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int px = (lane >> 4) * 4 + r;
            float v = (PP ? acc[i][j][r] * e.acc_scale : acc[i][j][r]) + bv[j];
            if (e.relu) v = fmaxf(v, 0.f);
            const TS q = (TS)v;
            *reinterpret_cast<TS*>(slab + px * SLAB_PITCH + (j * 16 + fr) * (int)sizeof(TS)) = q;
            const float vs = HPO ? storage_round<hp_t>(v) : (float)q;            // statistics of the value as STORED
            const bool ok = y < p.H && x0 + px < p.W;
            cs1[j] += ok ? vs : 0.f;
            cs2[j] += ok ? vs * vs : 0.f;
          }
        // (wave-private slab: LDS operations of one wave execute in order, no barrier)
        if constexpr (sizeof(TS) == 2) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int u = lane + k * 64, px = u >> 3, c8 = u & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + px * SLAB_PITCH + c8 * 16);
            if (y < p.H && x0 + px < p.W) {
              *reinterpret_cast<u32x4*>(outp + (((long)b * p.H + y) * p.W + x0 + px) * e.out_ld + c8 * 8) = v;
              if (SW && stats) {                  // (statistics of the values as STORED; c8 = lane & 7 for both k)
                float f8[8];
                load8(reinterpret_cast<const TS*>(&v), f8);
#pragma unroll
                for (int q = 0; q < 8; ++q) { s1[q] += f8[q]; s2[q] += f8[q] * f8[q]; }
              }
            }
          }
        } else {
          // 4-byte outputs: 16 lanes per pixel (256 bytes), one 16-byte store each: whole lines per wave-instruction; plane
          // pairs: a PAIR of lanes takes an 8-channel group, the even lane stores its hi plane, the odd lane its lo plane
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int u = lane + k * 64, px = u >> 4, c4 = u & 15;
            const bool ok = y < p.H && x0 + px < p.W;
            TO* dst = outp + (((long)b * p.H + y) * p.W + x0 + px) * e.out_ld + c4 * 4;
            if constexpr (HPO) {
              float v8[8];
              load8(reinterpret_cast<const float*>(slab + px * SLAB_PITCH) + (c4 >> 1) * 8, v8);
              u32x4 hi, lo;
              hp_split(v8, hi, lo);
              if (ok) *reinterpret_cast<u32x4*>(dst) = (c4 & 1) ? lo : hi;
            } else {
              const u32x4 v = *reinterpret_cast<const u32x4*>(slab + px * SLAB_PITCH + c4 * 16);
              if (ok) *reinterpret_cast<u32x4*>(dst) = v;
            }
          }
        }
      }
    }
  }
  if (stats) {
    if constexpr (SW) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float t1 = s1[q], t2 = s2[q];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (lane < 8) {
          atomicAdd(&sstat[lane * 8 + q], t1);
          atomicAdd(&sstat[BN + lane * 8 + q], t2);
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t1 = cs1[j], t2 = cs2[j];
      t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
      if (lane < 16) {
        atomicAdd(&sstat[j * 16 + lane], t1);
        atomicAdd(&sstat[BN + j * 16 + lane], t2);
      }
    }
    }
    __syncthreads();
    const long rep = (long)(blockIdx.x % (unsigned)e.stat_replicas) * e.N;
    if (tid < BN) {
      atomicAdd(&e.stat_sum[rep + tid], (double)sstat[tid]);
      atomicAdd(&e.stat_sumsq[rep + tid], (double)sstat[BN + tid]);
    }
  }
}

template <typename T16, typename TO = T16, bool PP = false>
int launch_c16(ConvParams p, hipStream_t st) {
  p.tiles_y = cdiv(p.H, TR);
  p.tiles_x = cdiv(p.W, TC);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  constexpr int OB = sizeof(TO) == 2 ? 2 : 4;               // bytes per staged output element
  constexpr size_t lds = (size_t)HALO_ROWS * 32 + 4 * 16 * (64 * OB + 16) + 3 * 64 * 4 + 9 * 64 * 32;   // 38.9 KB (47 KB)
  const long cap = OB == 2 ? 1024 : 768;                    // 4 (3) workgroups per CU
  const long grid = ntiles < cap ? ntiles : cap;
  hipLaunchKernelGGL((conv3x3_c16_kernel<T16, TO, PP>), dim3((unsigned)grid), dim3(256), lds, st, p, (int)ntiles);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// ---------------------------------------------------------------------------------------------------
// 64 -> 64 channels at 256x256 (level 0 / last decoder level, forward and input gradient: 4 launches per
// step): PERSISTENT kernel, one workgroup of 8 waves per CU.
//   * ALL weights (9 x 64 x 64 bf16 = 72 KB) are loaded into LDS once per workgroup: no weight DMA per tile
//     (64 % of the W4 kernel's LDS-DMA traffic), no weight ring, NO barrier inside the 9-tap loop;
//   * two teams of 4 waves, each with its own 16x16-pixel tile and halo buffer (2 x 40.5 KB); wave w of a team
//     owns image rows 4w .. 4w+3 x 64 channels (4 x 4 tiles of 16x16x32), as in the W4 kernel;
//   * the NEXT tile's halo travels global -> registers (11 x 16 B per thread) while the current tile is
//     computed and written, and moves registers -> LDS after it (the c16 kernel's scheme);
//   * output through wave-private slabs inside the (then dead) halo buffer, one 16-pixel image row (2 KB
//     contiguous) at a time; BatchNorm statistics (mode 1) and BatchNorm-backward sums (mode 2) stay in
//     registers across tiles and are flushed once per workgroup.
// Three barriers per tile (halo written | halo consumed | slabs consumed) instead of 9+.
// Round 1 (workgroup barriers, loads / stores under branches): 195 us vs 252-263 us (W4) at 64->64 @256x256, B = 32;
// MFMAs + loads alone 83 us, loads + output alone 110 us -- the two added up, which round 1 blamed on the power
// budget.  The phase stamps of round 2 (tools/diag_p64_phases.py) showed a compiler-inserted vmcnt(0) in front of
// the halo's register -> LDS move and the two teams in lock step; see below and DESIGN.md section 4.
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for every
// outstanding global STORE and prefetch load of the wave -- in a persistent kernel that drains the memory
// pipeline at every barrier and serialises the output stream with the MFMA phase.
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Every global access of the tile loop is a BUFFER load / store whose per-lane offset is out of range where the
// pixel is outside the image (loads return zeros, stores are dropped): no branch around any of them, so every wave
// issues the same number of vector-memory operations per tile and hipcc's wait-count pass can count them -- the
// wait in front of the halo's register -> LDS move becomes `s_waitcnt vmcnt(<stores of the previous tile>)`.  With
// the loads and stores under `if (pixel inside)` branches (the first form of this kernel) that wait was vmcnt(0):
// every tile waited for the previous tile's output stores to be ACKNOWLEDGED before its MFMA phase began, which is
// why the MFMA phase (83 us) and the memory phase (110 us) added up to the 195 us measured.  The prologue issues one
// all-out-of-range epilogue so that the loop entry and the back edge carry the same operation count (the pass
// merges the two states conservatively).
// Barrier of ONE team (4 of the workgroup's 8 waves): s_barrier counts all 8 waves, which keeps the two teams in the
// same phase -- both in their MFMA phase, then both writing out, the SIMDs idle meanwhile.  A counter in LDS instead:
// each wave adds 1 and polls until all four have (LDS executes one wave's operations in order, so a wave that sees
// the count has the others' earlier LDS writes and they have finished their reads).  The teams then run half a tile
// apart: one team's MFMA phase fills the SIMDs while the other moves data.
__device__ __forceinline__ void team_barrier(unsigned* cnt, unsigned& target, int lane) {
  target += 4;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

template <int MODE, bool POOL, typename T16>   // epilogue statistics (0 none, 1 BatchNorm statistics, 2 BatchNorm-backward sums); fused 2x2 max-pool
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_p64_kernel(ConvParams p, int ntiles) {
  static_assert(!POOL || MODE == 0, "the fused max-pool is an inference epilogue");
#ifdef CRIMAC_DIAG_PHASES
  unsigned long long rt[4];
#define CRIMAC_RT(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
  CRIMAC_RT(0)
#else
#define CRIMAC_RT(k)
#endif
  constexpr int BN = 64, NT = 4;
  constexpr int W_BYTES = 9 * BN * RB;               // 73728
  constexpr int H_BYTES = HALO_ROWS * RB;            // 41472
  constexpr int NHU = (HALO_ROWS * 8 + 255) / 256;   // 16-byte halo units per thread: 11
  constexpr int SLAB_PITCH = BN * 2 + 16, SLAB = 16 * SLAB_PITCH;
  constexpr unsigned OOB = 0x80000000u;              // >= num_records of the rebased buffer resources
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;
  const int tid = threadIdx.x, tt = tid & 255, lane = tid & 63;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane(tt >> 6);
  unsigned char* halo = smem + W_BYTES + team * H_BYTES;
  const EpiParams& e = p.epi;
  constexpr int NSL = 2;                                   // wave-private slabs: image rows 2k, 2k + 1 (the fused max-pool
  unsigned char* slab0 = halo + wave * NSL * SLAB;         // needs both rows of a window)
  static_assert(4 * NSL * SLAB <= H_BYTES, "the slabs live in the team's (dead) halo buffer");
  float* sstat = reinterpret_cast<float*>(smem + W_BYTES + 2 * H_BYTES);
  // The MFMAs take the WEIGHT fragment as the A operand (round 5, as the first-layer kernel): a lane holds four consecutive
  // CHANNELS of one pixel, the tile is staged with one packed 8-byte LDS write per accumulator tile instead of four conversions
  // and four 2-byte writes (a quarter of the staging work made the plain launches 11-14 % faster in the ablation build).  The
  // bias comes from LDS (four consecutive floats per accumulator tile) and the mode-1 statistics are taken where the slab is read back, like mode 2's sums.
  float* sbias = sstat + 2 * BN + 8;                       // [64], behind the statistics and the team counters
  const T16* inp = reinterpret_cast<const T16*>(p.in);
  T16* outp = reinterpret_cast<T16*>(e.out);
  constexpr int mode = MODE;
  if (tid < 2 * BN) sstat[tid] = 0.f;
  if (tid < BN) sbias[tid] = e.bias ? e.bias[p.n_first + tid] : 0.f;

  // weights: LDS unit c of row (t, n) holds source unit c ^ swizzle(n) (as the W4 weight slots)
  {
    static_assert(9 * BN * 8 == 9 * 512, "nine 16-byte units per thread");
    u32x4 wreg[9];                               // (all nine requests in flight, one wait: the prologue is latency)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int u = tid + k * 512;
      const int row = u >> 3, c = u & 7, n = row & (BN - 1);
      const int t = row >> 6;                    // (rows of a range: tap stride N * Cin, first row n_first)
      wreg[k] = *reinterpret_cast<const u32x4*>(p.w_hi + ((long)t * p.N + p.n_first + n) * BN + ((c ^ ((n >> 1) & 7)) * 8));
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) *reinterpret_cast<u32x4*>(wl + (tid + k * 512) * 16) = wreg[k];
  }

  const int fr = lane & 15, fq = lane >> 4;
  const int bsw = (fr >> 1) & 7;                     // weight-row swizzle of rows j*16 + fr
  int asw[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) asw[kx] = halo_swz(fr + kx);
  // mode 2: this lane always stores channel chunk c8 = lane & 7 -> its BatchNorm constants live in registers
  const int c8 = lane & 7;
  float sc[8], sh[8], mu[8], d1[8], d2[8];        // (d2 accumulates dz * (y - mean); invstd is applied at the flush)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = p.n_first + c8 * 8 + k;
    mu[k] = mode == 2 ? e.bnb_vec[c] : 0.f;
    sc[k] = mode == 2 ? e.bnb_vec[2 * e.bnb_stride + c] : 0.f;
    sh[k] = mode == 2 ? e.bnb_vec[3 * e.bnb_stride + c] : 0.f;
    d1[k] = 0.f;
    d2[k] = 0.f;
  }

  auto tile_geo = [&](int tile, int& b, int& y0, int& x0) {
    const int txi = tile % p.tiles_x;
    const int q = tile / p.tiles_x;
    b = q / p.tiles_y;
    y0 = (q % p.tiles_y) * TR;
    x0 = txi * TC;
  };
  // this lane's halo units: byte offset from the halo's first pixel (y0 - 1, x0 - 1) and its (row, column) there
  // (H and W are multiples of the tile size here -- launch_p64 -- so a halo pixel is outside the image only on the
  // tile's first / last row or column facing an image border: four flag bits, kept in the idle low bits of the offset)
  unsigned h_rel[NHU];
#pragma unroll
  for (int i = 0; i < NHU; ++i) {
    const int u = tt + i * 256;
    const int row = u >> 3, c = u & 7;
    const int hy = row / HP, hx = row - hy * HP;
    const unsigned off = (unsigned)((((long)hy * p.W + hx) * p.in_ld + ((c ^ halo_swz(hx)) * 8)) * 2);     // % 16 == 0
    const unsigned edge = (hy == 0 ? 1u : 0u) | (hx == 0 ? 2u : 0u) | (hy == TR + 1 ? 4u : 0u) | (hx == TC + 1 ? 8u : 0u);
    h_rel[i] = u < HALO_ROWS * 8 ? (off | edge) : (OOB | 15u);
  }
  u32x4 hreg[NHU];
  auto fetch_halo = [&](int tile, bool valid, int i0 = 0, int i1 = NHU) {      // always the same loads per lane
    int b, y0, x0;
    tile_geo(valid ? tile : 0, b, y0, x0);
    // image borders this tile touches (no tile: all four, and the out-of-range bit of every offset)
    const unsigned border = valid ? ((y0 == 0 ? 1u : 0u) | (x0 == 0 ? 2u : 0u) | (y0 + TR >= p.H ? 4u : 0u) |
                                     (x0 + TC >= p.W ? 8u : 0u)) : 15u;
    const unsigned kill = valid ? 0u : OOB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T16*>(inp + (((long)b * p.H + y0 - 1) * p.W + x0 - 1) * p.in_ld), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NHU; ++i)
      if (i >= i0 && i < i1)
        hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(
            rs, (int)((h_rel[i] & border) ? OOB : ((h_rel[i] & ~15u) | kill)), 0, CRIMAC_P64_HALO_AUX);
  };

  // this lane's output units: rows wave * 4 + i, pixels (lane >> 3) + 8 k, channel chunk c8
  const long out_row = (long)p.W * e.out_ld * 2, y_row = mode == 2 ? (long)p.W * e.bnb_y_ld * 2 : 0;
  const unsigned o_rel = (unsigned)((wave * 4) * out_row + (lane >> 3) * e.out_ld * 2 + c8 * 16);
  const unsigned y_rel = mode == 2 ? (unsigned)((wave * 4) * y_row + (lane >> 3) * e.bnb_y_ld * 2 + c8 * 16) : 0u;
  const int Hp = p.H >> 1, Wp = p.W >> 1;
  const unsigned q_rel = POOL ? (unsigned)(((long)(wave * 2) * Wp + (lane >> 3)) * e.pool_ld * 2 + c8 * 16) : 0u;

  // writes one tile (accumulators -> bias / ReLU -> wave-private slabs -> 16-byte stores) and folds the statistics;
  // valid == false: the same instruction stream with every store and load out of range
  auto epilogue = [&](const f32x4 (&acc)[4][NT], bool valid, int b, int y0, int x0) {
    const int Hl = valid ? p.H : 0;                      // (as in fetch_halo)
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        outp + (((long)b * p.H + y0) * p.W + x0) * e.out_ld + p.n_first, 0, 0x7FFFFFFF, 0x00020000);
    // mode 2: the saved forward outputs of image row i, requested one row ahead and BEFORE that row's stores, so
    // that their wait leaves the youngest stores in flight.  (All four rows up front -- 32 registers -- would leave
    // the whole tile's stores in flight, but hipcc then spills the prefetched halo into scratch, i.e. into vmcnt.)
    u32x4 yreg[2][2];
    auto load_y = [&](int i) {
      if constexpr (mode == 2) {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T16*>(reinterpret_cast<const T16*>(e.bnb_y) + (((long)b * p.H + y0) * p.W + x0) * e.bnb_y_ld +
                             p.n_first), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool ok = (y0 + wave * 4 + i < Hl) & (x0 + (lane >> 3) + 8 * k < p.W);
          yreg[i & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(
              ry, (int)(ok ? y_rel + (unsigned)(i * y_row + 8 * k * e.bnb_y_ld * 2) : OOB), 0, 0);
        }
      }
    };
    auto to_slab = [&](int i) {                  // bias / ReLU / rounding, one image row -> its slab
      unsigned char* slab = slab0 + (i % NSL) * SLAB;
      // accumulator element r of N tile j: pixel fr of the row, channel j*16 + fq*4 + r (bias: four consecutive floats in LDS,
      // added to the finished sum like every other convolution kernel does -- 64-channel ranges equal the full convolution bit for bit)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + j * 16 + fq * 4);
        T16 q4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef CRIMAC_EXP_P64_FEWEPI      // (ablation build, results garbage: one of the four staging iterations per accumulator tile)
          if (r != 0) continue;
#endif
          float v = acc[i][j][r] + b4[r];
          if (e.relu) v = fmaxf(v, 0.f);
          q4[r] = (T16)v;
        }
        *reinterpret_cast<u32x2*>(slab + fr * SLAB_PITCH + (j * 16 + fq * 4) * 2) = *reinterpret_cast<const u32x2*>(q4);
      }
    };
    auto from_slab = [&](int i) {                // (wave-private slab: the LDS operations of one wave execute in order)
      const int y = y0 + wave * 4 + i;
      unsigned char* slab = slab0 + (i % NSL) * SLAB;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = (lane >> 3) + 8 * k;
        const T16* sp = reinterpret_cast<const T16*>(slab + px * SLAB_PITCH) + c8 * 8;
        const bool ok = (y < Hl) & (x0 + px < p.W);
        const u32x4 val = *reinterpret_cast<const u32x4*>(sp);
        __builtin_amdgcn_raw_buffer_store_b128(val, ro, (int)(ok ? o_rel + (unsigned)(i * out_row + 8 * k * e.out_ld * 2) : OOB), 0, 0);
        if constexpr (mode == 1) {               // statistics of the values as STORED, this lane's channel chunk c8
          float g[8];
          load8(sp, g);
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            d1[kk] += ok ? g[kk] : 0.f;
            d2[kk] += ok ? g[kk] * g[kk] : 0.f;
          }
        }
        if constexpr (mode == 2) {
          // This epilogue sits at the 256-register limit and hipcc's allocation around it is erratic (spill counts
          // between 3 and 220 for equivalent formulations).  Two forms, each kept for the storage type where it
          // compiles without heavy spilling AND passes test_conv3x3_dgrad_with_fused_bn_backward_sums: unpacking two
          // channels at a time from the packed registers (fp16: 60 -> 18 spilled registers, 393 -> 251 us; the same
          // source gave WRONG sums for channels 2 mod 8 in the bf16 build), and unpacking both vectors first (bf16).
          if constexpr (sizeof(T16) == 2 && __is_same(T16, half_t)) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              const unsigned gw = val[h], yw = yreg[i & 1][k][h];
#pragma unroll
              for (int q2 = 0; q2 < 2; ++q2) {
                const int kk = 2 * h + q2;
                const float gk = E16<T16>::val((unsigned short)(q2 ? gw >> 16 : gw & 0xffffu));
                const float yk = E16<T16>::val((unsigned short)(q2 ? yw >> 16 : yw & 0xffffu));
                const float dz = (ok && (yk * sc[kk] + sh[kk]) > 0.f) ? gk : 0.f;
                d1[kk] += dz;
                d2[kk] += dz * (yk - mu[kk]);
              }
            }
          } else {
            float g[8], yv[8];
            load8(sp, g);
            load8(reinterpret_cast<const T16*>(&yreg[i & 1][k]), yv);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const float dz = (ok && (yv[kk] * sc[kk] + sh[kk]) > 0.f) ? g[kk] : 0.f;
              d1[kk] += dz;
              d2[kk] += dz * (yv[kk] - mu[kk]);
            }
          }
        }
      }
      if constexpr (POOL) {
        if (i & 1) {
          // rows y - 1 and y are in the two slabs: pooled row (y >> 1), 8 pooled pixels x 8 channel chunks = 64 lanes
          const int pxl = lane >> 3;
          const bool ok = (y < Hl) & ((x0 >> 1) + pxl < Wp);
          float m[8], v[8];
          load8(reinterpret_cast<const T16*>(slab0 + (2 * pxl) * SLAB_PITCH) + c8 * 8, m);
#pragma unroll
          for (int d = 1; d < 4; ++d) {
            load8(reinterpret_cast<const T16*>(slab0 + (d >> 1) * SLAB + (2 * pxl + (d & 1)) * SLAB_PITCH) + c8 * 8, v);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) m[kk] = fmaxf(m[kk], v[kk]);
          }
          u32x4 pv;
          store8(reinterpret_cast<T16*>(&pv), m);
          const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
              reinterpret_cast<T16*>(e.pool_out) + (((long)b * Hp + (y0 >> 1)) * Wp + (x0 >> 1)) * e.pool_ld + p.n_first,
              0, 0x7FFFFFFF, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(pv, rq, (int)(ok ? q_rel + (unsigned)((long)(i >> 1) * Wp * e.pool_ld * 2) : OOB), 0, 0);
        }
      }
    };
    load_y(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      to_slab(i);
      if (i + 1 < 4) load_y(i + 1);
      from_slab(i);
    }
  };

  // tiles of this workgroup: blockIdx.x + j * gridDim.x; the teams take the next j from a queue counter in LDS (the
  // team whose waves are older on their SIMDs runs ~20 % faster: with a fixed half each it idled through the tail)
  const int stride = gridDim.x;
  int tile = blockIdx.x + team * stride;
  fetch_halo(tile, tile < ntiles);
  {
    f32x4 zacc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) zacc[i][j][r] = 0.f;
    epilogue(zacc, false, 0, 0, 0);             // (see above: equal operation counts on both edges into the loop)
  }
  // the teams loop independently (team barriers); team 1 starts half a tile late
  unsigned* tq = reinterpret_cast<unsigned*>(sstat + 2 * BN);      // [0..1] team barriers, [2] tile queue, [4..5] next j
  unsigned* tcnt = tq + team;
  unsigned ttarget = 0;
  if (tid < 3) tq[tid] = tid == 2 ? 2u : 0u;
  wg_barrier_lds();                             // weights, statistics and counters in place
  if (team == 1) {
#pragma unroll 1
    for (int k = 0; k < 3; ++k) __builtin_amdgcn_s_sleep(32);
  }
  CRIMAC_DIAG_STAMP(dg_t0, dg_r0)
  CRIMAC_RT(1)
#ifdef CRIMAC_DIAG_PHASES
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, ph_t = dg_t0;
#define CRIMAC_PH(k) { unsigned long long tn; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[k] += tn - ph_t; ph_t = tn; }
  int niter = 0;
#else
#define CRIMAC_PH(k)
#endif
  while (tile < ntiles) {
    int b, y0, x0;
    tile_geo(tile, b, y0, x0);
    team_barrier(tcnt, ttarget, lane);          // slabs of the previous tile consumed
    CRIMAC_PH(0)
    if (tt == 0) tq[4 + team] = __hip_atomic_fetch_add(tq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int i = 0; i < NHU; ++i) {
      const int u = tt + i * 256;
      if (u < HALO_ROWS * 8) *reinterpret_cast<u32x4*>(halo + u * 16) = hreg[i];
    }
    team_barrier(tcnt, ttarget, lane);          // halo in place
    CRIMAC_PH(1)

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // the next tile's halo is requested in three parts, one in front of each tap row: a load that has to wait for
    // room behind the previous tile's stores then stalls this wave while the other team's wave on the SIMD computes
    // (tap rows not unrolled into one body: the compiler then hoists fragment reads until registers spill)
    auto tap_row = [&](int ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int t = ky * 3 + kx;
        const unsigned char* a0 = halo + ((wave * 4 + ky) * HP + fr + kx) * RB;
        const unsigned char* b0 = wl + (t * BN + fr) * RB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int unit = 4 * ks + fq;
          bf16x8 af[4], bfr[NT];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(a0 + i * (HP * RB) + ((unit ^ asw[kx]) << 4));
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bfr[j] = *reinterpret_cast<const bf16x8*>(b0 + j * (16 * RB) + ((unit ^ bsw) << 4));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = E16<T16>::mfma16(bfr[j], af[i], acc[i][j]);      // (weights as A: rows = channels, lane column = pixel)
        }
      }
    };
    // (written by the team's first wave in front of the barrier above)
    const int next = (int)blockIdx.x + __builtin_amdgcn_readfirstlane((int)*(volatile unsigned*)(tq + 4 + team)) * stride;
    const bool more = next < ntiles;
    fetch_halo(next, more, 0, 4);
    tap_row(0);
    __builtin_amdgcn_sched_barrier(0);
    fetch_halo(next, more, 4, 8);
    tap_row(1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_halo(next, more, 8, NHU);
    tap_row(2);
    team_barrier(tcnt, ttarget, lane);          // halo consumed: its buffer now holds the output slabs
    CRIMAC_PH(2)
    epilogue(acc, true, b, y0, x0);
    CRIMAC_PH(3)
#ifdef CRIMAC_DIAG_PHASES
    ++niter;
#endif
    tile = next;
  }
  CRIMAC_DIAG_STAMP(dg_t1, dg_r1)
#ifdef CRIMAC_DIAG_PHASES
  // cycles of wave 0 of each team per phase: wait for the slabs | halo registers -> LDS (incl. the wait for the
  // prefetch) | prefetch issue + MFMA phase | epilogue
  CRIMAC_RT(2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last stores acknowledged)
  CRIMAC_RT(3)
  if (lane == 0 && wave == 0)
    for (int k = 0; k < 4; ++k) crimac_diag_clock_conv_buf[5120 + (blockIdx.x * 2 + team) * 4 + k] = rt[k];
  if (lane == 0 && wave == 0)
    for (int k = 0; k < 5; ++k) {
      crimac_diag_clock_conv_buf[2 * ((blockIdx.x * 2 + team) * 5 + k)] = ph[k];
      crimac_diag_clock_conv_buf[2 * ((blockIdx.x * 2 + team) * 5 + k) + 1] = (unsigned long long)niter;
    }
#else
  CRIMAC_DIAG_STORE(crimac_diag_clock_conv, dg_t0, dg_r0, dg_t1, dg_r1)
#endif
  if (mode == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        d1[k] += __shfl_xor(d1[k], o, 64);
        d2[k] += __shfl_xor(d2[k], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        atomicAdd(&sstat[c8 * 8 + k], d1[k]);
        atomicAdd(&sstat[BN + c8 * 8 + k], d2[k]);
      }
    }
  } else if (mode == 2) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        d1[k] += __shfl_xor(d1[k], o, 64);
        d2[k] += __shfl_xor(d2[k], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        atomicAdd(&sstat[c8 * 8 + k], d1[k]);
        atomicAdd(&sstat[BN + c8 * 8 + k], d2[k] * e.bnb_vec[e.bnb_stride + p.n_first + c8 * 8 + k]);
      }
    }
  }
  if (mode) {
    __syncthreads();
    const long rep = (long)(blockIdx.x % (unsigned)e.stat_replicas) * e.N;
    if (tid < BN) {
      atomicAdd(&e.stat_sum[rep + p.n_first + tid], (double)sstat[tid]);
      atomicAdd(&e.stat_sumsq[rep + p.n_first + tid], (double)sstat[BN + tid]);
    }
  }
}

template <typename T16>
int launch_p64(ConvParams p, hipStream_t st) {
  p.tiles_y = cdiv(p.H, TR);
  p.tiles_x = cdiv(p.W, TC);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  constexpr size_t lds = (size_t)9 * 64 * RB + 2 * (size_t)HALO_ROWS * RB + 2 * 64 * 4 + 32 + 64 * 4;     // 157472 B (statistics, team counters, bias)
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p64_kernel<0, false, T16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p64_kernel<0, true, T16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p64_kernel<1, false, T16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p64_kernel<2, false, T16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  CRIMAC_REQUIRE(p.H % TR == 0 && p.W % TC == 0, "conv3x3 (64 -> 64 persistent kernel): H, W must be multiples of 16");
  const int ncu = crimac_cu_count();
  long grid = (ntiles + 1) / 2;
  if (grid > ncu) grid = ncu;
  const int mode = p.epi.stat_sum ? p.epi.stat_mode : 0;
  if (mode == 0 && p.epi.pool_out)
    hipLaunchKernelGGL((conv3x3_p64_kernel<0, true, T16>), dim3((unsigned)grid), dim3(512), lds, st, p, (int)ntiles);
  else if (mode == 0)
    hipLaunchKernelGGL((conv3x3_p64_kernel<0, false, T16>), dim3((unsigned)grid), dim3(512), lds, st, p, (int)ntiles);
  else if (mode == 1)
    hipLaunchKernelGGL((conv3x3_p64_kernel<1, false, T16>), dim3((unsigned)grid), dim3(512), lds, st, p, (int)ntiles);
  else
    hipLaunchKernelGGL((conv3x3_p64_kernel<2, false, T16>), dim3((unsigned)grid), dim3(512), lds, st, p, (int)ntiles);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

}  // namespace

// first layer: 16-bit storage (fp16 != 0: IEEE half, else bf16), Cin == 16 (4 real channels), N == 64
int crimac_conv3x3_c16_16(const void* in, long in_ld, int B, int H, int W, int N, const void* w_hi,
                          const EpiParams& epi, hipStream_t st, int fp16) {
  ConvParams p;
  p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.Cin = 16; p.N = N;
  p.w_hi = (const unsigned short*)w_hi;
  p.epi = epi;
  p.n_first = 0; p.n_count = N;
  return fp16 ? launch_c16<half_t>(p, st) : launch_c16<bf16_t>(p, st);
}

// first layer, plane pairs (CRIMAC_PREC_H3P): in = [pixels][in_ld channels] of hp_t (in_ld >= 8), w = interleaved planes
// [9][N][16 hi | 16 lo]; output fp32 or (out_planes) plane pairs
int crimac_conv3x3_c16_hp(const void* in, long in_ld, int B, int H, int W, int N, const void* w, const EpiParams& epi,
                          hipStream_t st, int out_planes) {
  ConvParams p;
  p.in = in; p.in_ld = 2 * in_ld; p.B = B; p.H = H; p.W = W; p.Cin = 16; p.N = N;
  p.w_hi = (const unsigned short*)w;
  p.epi = epi;
  p.n_first = 0; p.n_count = N;
  return out_planes ? launch_c16<half_t, hp_t, true>(p, st) : launch_c16<half_t, float, true>(p, st);
}

namespace {
template <typename T16>
int glds_dispatch(ConvParams p, hipStream_t st) {
  const int B = p.B, H = p.H, W = p.W, Cin = p.Cin, N = p.N, n_first = p.n_first, n_count = p.n_count;
  const long in_ld = p.in_ld;
  if (p.wfrag) {
    // a fragment-major weight plane: only the channel-split kernel reads it (128-channel form; CRIMAC_EPI_WROWS: rows form)
    const bool small_f = (((long)B * H * W - 1) * in_ld + Cin) * 2 < (1L << 31);
    if (p.wfrag == 2) {
      CRIMAC_REQUIRE(N % 64 == 0 && n_first % 64 == 0 && n_count % 64 == 0 && Cin % 64 == 0 && small_f,
                     "conv3x3: the rows form (CRIMAC_EPI_WROWS) needs N and the channel range in multiples of 64, Cin %% 64 == 0 "
                     "and an input tensor below 2 GB (N=%d range [%d, +%d) Cin=%d)", N, n_first, n_count, Cin);
      return launch_wch<T16, T16, false, 3, true>(p, st);
    }
    CRIMAC_REQUIRE(N % 128 == 0 && n_first % 128 == 0 && n_count % 128 == 0 && Cin % 64 == 0,
                   "conv3x3: fragment-major weights (CRIMAC_EPI_WFRAG) need N and the channel range in multiples of 128 and "
                   "Cin %% 64 == 0 (N=%d range [%d, +%d) Cin=%d)", N, n_first, n_count, Cin);
    CRIMAC_REQUIRE(small_f, "conv3x3: fragment-major weights (CRIMAC_EPI_WFRAG): the input tensor exceeds the 2 GB the "
                   "channel-split kernel addresses (B=%d H=%d W=%d ld=%ld): pack row-major planes (CRIMAC_WFRAG=0)", B, H, W, in_ld);
    return launch_wch<T16, T16, false, 0, true>(p, st);
  }
  if (n_first != 0 || n_count != N) {
    // a range of output channels (crimac_conv3x3_cols): 64 of them with the persistent 64-channel kernel,
    // multiples of 128 with the channel-split kernel
    const long nt = (long)B * cdiv(H, TR) * cdiv(W, TC);
    // (the persistent kernel takes its statistics from the STORED values: unrounded sums -- CRIMAC_EPI_STAT_RAW -- go to the tall form)
    if (n_count == 64 && Cin == 64 && nt >= 512 && H % TR == 0 && W % TC == 0 && !p.epi.stat_raw) return launch_p64<T16>(p, st);
    const bool small_t = (((long)B * H * W - 1) * in_ld + Cin) * 2 < (1L << 31);
    // other 64-channel ranges (small images, Cin != 64): the tall form of the channel-split kernel
    if ((n_first % 128 != 0 || n_count % 128 != 0) && n_first % 64 == 0 && n_count % 64 == 0 && small_t)
      return launch_wch<T16, T16, false, 2>(p, st);
    CRIMAC_REQUIRE(n_first % 128 == 0 && n_count % 128 == 0 && small_t,
                   "conv3x3_cols: channel range [%d, +%d) of %d not supported (multiples of 128, or 64 of a "
                   "64-input-channel convolution with >= 512 tiles)", n_first, n_count, N);
    return launch_wch<T16>(p, st);
  }
  // N % 128 == 0: channel-split kernel; N = 64 (or 192, ...): pixel-split kernel with the LDS weight ring.
  // Measured per layer at B = 32 (tools/bench_conv.py): wch 1.1-1.6 PFLOP/s vs 1.0-1.25 (W4<128>); on the two
  // HBM-heavy N = 64 shapes W4<64> (305 / 423 us) beats a 2x2-wave channel split (319 / 458 us).
  static const int w4 = getenv("CRIMAC_CONV_W4") ? atoi(getenv("CRIMAC_CONV_W4")) : 0;
  // N = 64: the tall form of the channel-split kernel (CRIMAC_CONV_TALL16: 0 off, 1 every shape but 64 -> 64, 2 every N = 64
  // shape).  Measured at B = 32, 256 x 256: 128 -> 64  274-276 us (pixel-split kernel with the LDS weight ring: 373-379);
  // 64 -> 64  231-238 us against 175-190 of the persistent kernel below, which keeps that shape.
  static const int tall = getenv("CRIMAC_CONV_TALL16") ? atoi(getenv("CRIMAC_CONV_TALL16")) : 1;
  const bool small64 = (((long)B * H * W - 1) * in_ld + Cin) * 2 < (1L << 31);
  if (N == 64 && small64 && (tall == 2 || (tall == 1 && Cin != 64))) return launch_wch<T16, T16, false, 2>(p, st);
  // 64 -> 64 with many tiles: persistent kernel with LDS-resident weights (CRIMAC_CONV_P64=0: W4 for A/B runs)
  static const int p64 = getenv("CRIMAC_CONV_P64") ? atoi(getenv("CRIMAC_CONV_P64")) : 1;
  if (p64 && N == 64 && Cin == 64 && H % TR == 0 && W % TC == 0 && (long)B * (H / TR) * (W / TC) >= 512 && !p.epi.stat_raw)
    return launch_p64<T16>(p, st);
  if (N % 128 != 0) return launch_w4<64, T16>(p, st);
  const bool small = (((long)B * H * W - 1) * in_ld + Cin) * 2 < (1L << 31);     // 32-bit buffer offsets in wch
  return (w4 == 1 || !small) ? launch_w4<128, T16>(p, st) : launch_wch<T16>(p, st);
}
}  // namespace

// Plane-pair input (CRIMAC_PREC_H3P), Cin % 32 == 0, N % 64 == 0: the 16-bit kernels on a tensor of 2 Cin halves per
// pixel (3 MFMAs per product); output fp32 (out_planes == 0) or plane pairs.  The persistent 64 -> 64 kernel has no
// plane-pair form (144 KB of weights do not fit beside the halos): those layers take the pixel-split kernel.
int crimac_conv3x3_glds_hp(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w,
                           const EpiParams& epi, hipStream_t st, int n_first, int n_count, int out_planes, int wfrag) {
  ConvParams p;
  p.in = in; p.in_ld = 2 * in_ld; p.B = B; p.H = H; p.W = W; p.Cin = 2 * Cin; p.N = N;
  p.w_hi = (const unsigned short*)w;
  p.epi = epi;
  p.n_first = n_first; p.n_count = n_count;
  const bool small = (((long)B * H * W - 1) * p.in_ld + p.Cin) * 2 < (1L << 31);     // 32-bit buffer offsets in wch
  static const int w4 = getenv("CRIMAC_CONV_W4") ? atoi(getenv("CRIMAC_CONV_W4")) : 0;
  CRIMAC_REQUIRE(wfrag != 2, "conv3x3 (plane pairs): no rows form (CRIMAC_EPI_WROWS)");
  if (wfrag) {      // fragment-major plane (rows of 2 Cin halves): the 128-channel form of the channel-split kernel only
    CRIMAC_REQUIRE(N % 128 == 0 && n_count % 128 == 0 && n_first % 128 == 0 && small,
                   "conv3x3 (plane pairs): fragment-major weights (CRIMAC_EPI_WFRAG) need N and the channel range in multiples of "
                   "128 and an input tensor below 2 GB (N=%d range [%d, +%d))", N, n_first, n_count);
    return out_planes ? launch_wch<half_t, hp_t, true, 0, true>(p, st) : launch_wch<half_t, float, true, 0, true>(p, st);
  }
  if (n_count % 128 == 0 && n_first % 128 == 0 && small && w4 != 1)
    return out_planes ? launch_wch<half_t, hp_t, true>(p, st) : launch_wch<half_t, float, true>(p, st);
  CRIMAC_REQUIRE(n_first % 64 == 0 && n_count % 64 == 0, "conv3x3 (plane pairs): channel range [%d, +%d) must be "
                 "multiples of 64", n_first, n_count);
  if (n_count % 128 == 0 && n_first % 128 == 0)
    return out_planes ? launch_w4<128, half_t, hp_t, true>(p, st) : launch_w4<128, half_t, float, true>(p, st);
  // 64-channel ranges: CRIMAC_CONV_S22 = 2 (default) the tall 32 x 16-pixel form, 1 the 2 x 2 form on 16 x 16 pixels,
  // 0 the pixel-split kernel with the LDS weight ring
  static const int s22 = getenv("CRIMAC_CONV_S22") ? atoi(getenv("CRIMAC_CONV_S22")) : 2;
  if (s22 == 2 && small)
    return out_planes ? launch_wch<half_t, hp_t, true, 2>(p, st) : launch_wch<half_t, float, true, 2>(p, st);
  if (s22 && small)
    return out_planes ? launch_wch<half_t, hp_t, true, 1>(p, st) : launch_wch<half_t, float, true, 1>(p, st);
  return out_planes ? launch_w4<64, half_t, hp_t, true>(p, st) : launch_w4<64, half_t, float, true>(p, st);
}

// CRIMAC_PREC_H3F_BWD: fp16 input and weights (1 MFMA per product), fp32 output -- the input-gradient convolutions of the
// 'h3f' mode (their output `da` feeds elementwise kernels that decide ReLU masks and pool positions on fp32 values, and the
// fused BatchNorm-backward sums read the forward pass's fp32 y).  The same kernel forms as the 16-bit modes; 64-channel
// ranges take the tall form (the persistent 64 -> 64 kernel has no fp32 output).
int crimac_conv3x3_glds_16_f32out(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w_hi,
                                  const EpiParams& epi, hipStream_t st, int n_first, int n_count, int wfrag) {
  ConvParams p;
  p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.w_hi = (const unsigned short*)w_hi;
  p.epi = epi;
  p.n_first = n_first; p.n_count = n_count;
  const bool small = (((long)B * H * W - 1) * in_ld + Cin) * 2 < (1L << 31);     // 32-bit buffer offsets in wch
  if (wfrag == 2) {
    CRIMAC_REQUIRE(N % 64 == 0 && n_count % 64 == 0 && n_first % 64 == 0 && small,
                   "conv3x3 (fp16 operands, fp32 output): the rows form (CRIMAC_EPI_WROWS) needs N and the channel range in "
                   "multiples of 64 and an input tensor below 2 GB (N=%d range [%d, +%d))", N, n_first, n_count);
    return launch_wch<half_t, float, false, 3, true>(p, st);
  }
  if (wfrag) {
    CRIMAC_REQUIRE(N % 128 == 0 && n_count % 128 == 0 && n_first % 128 == 0 && small,
                   "conv3x3 (fp16 operands, fp32 output): fragment-major weights (CRIMAC_EPI_WFRAG) need N and the channel range in "
                   "multiples of 128 and an input tensor below 2 GB (N=%d range [%d, +%d))", N, n_first, n_count);
    return launch_wch<half_t, float, false, 0, true>(p, st);
  }
  if (n_count % 128 == 0 && n_first % 128 == 0)
    return small ? launch_wch<half_t, float, false>(p, st) : launch_w4<128, half_t, float, false>(p, st);
  CRIMAC_REQUIRE(n_first % 64 == 0 && n_count % 64 == 0, "conv3x3 (fp16 operands, fp32 output): channel range [%d, +%d) must be "
                 "multiples of 64", n_first, n_count);
  return small ? launch_wch<half_t, float, false, 2>(p, st) : launch_w4<64, half_t, float, false>(p, st);
}

// 16-bit storage, Cin % 64 == 0, N % 64 == 0; argument checks are done by crimac_conv3x3 (conv3x3.hip).
int crimac_conv3x3_glds_16(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w_hi,
                           const EpiParams& epi, hipStream_t st, int n_first, int n_count, int fp16, int wfrag) {
  ConvParams p;
  p.wfrag = wfrag;
  p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.w_hi = (const unsigned short*)w_hi;
  p.epi = epi;
  p.n_first = n_first; p.n_count = n_count;
  return fp16 ? glds_dispatch<half_t>(p, st) : glds_dispatch<bf16_t>(p, st);
}
