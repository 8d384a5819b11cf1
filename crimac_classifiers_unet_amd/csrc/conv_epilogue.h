// Shared epilogue of the 3x3 convolution kernels (conv3x3.hip, conv3x3_glds.hip).
//
//   bias (+ReLU) in registers -> tile staged through LDS -> coalesced 16-byte stores, plus one of
//   the fused per-channel reductions (fp64 atomics spread over `stat_replicas` accumulators):
//     stat_mode 1: sum / sum of squares of the STORED output (BatchNorm batch statistics of the
//                  layer the convolution feeds, unet.py:78,81,121-122);
//     stat_mode 2: BatchNorm+ReLU BACKWARD sums of the layer the (input-gradient) convolution feeds:
//                  with da = this output, y = that layer's saved conv output, dz = da * [y*scale+shift > 0],
//                  xhat = (y-mean)*invstd:  stat_sum += sum dz, stat_sumsq += sum dz*xhat
//                  -- the bn_bwd_reduce pass over (da, y) is fused away.
#pragma once
#include <type_traits>

#include "common.h"

struct EpiParams {
  const float* bias;
  void* out;
  long out_ld;
  int relu, H, W, N;
  double* stat_sum;
  double* stat_sumsq;
  int stat_replicas;
  int stat_mode;
  const void* bnb_y;      // [pixels][bnb_y_ld], same pixel grid as the output
  long bnb_y_ld;
  const float* bnb_vec;   // rows: mean, invstd, scale, shift; row stride bnb_stride
  long bnb_stride;
  float acc_scale;        // the accumulators are multiplied by this before the bias (F32H3: 2^-WSHIFT; else 1)
  void* pool_out;         // != NULL: also the 2x2/2 max-pool of the stored tile, [B][H/2][W/2][pool_ld] (eval path:
  long pool_ld;           //          nn.MaxPool2d behind the block, unet.py:85-86 -- no second pass over the output)
  int stat_raw;           // stat_mode 1: sum the fp32 results BEFORE the rounding of the store (CRIMAC_EPI_STAT_RAW: the
                          // sums are a bias gradient, not the statistics of a stored tensor)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0) -- here that means for the
// ACKNOWLEDGEMENT of the tile's global stores: between the two slices of an fp32 tile (and in front of the statistics
// flush) every wave of the workgroup sat through a store round trip.  What the staging tile needs is that every wave
// has READ its part (its ds_reads have returned: the stores that consume them have issued); the stores themselves may
// stay in flight.  (Phase stamps, 64 -> 64 @256x256 plane pairs: epilogue 16.2 k cycles per workgroup next to 13.8 k of
// MFMA work.)
__device__ __forceinline__ void epi_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// -DCRIMAC_DIAG_EPI (diagnostic builds of ONE translation unit, tools/diag_epi_phases.py): s_memtime at the phase
// boundaries of the epilogue, wave 0 of each workgroup -> crimac_epi_buf[workgroup % 1024][8]
#ifdef CRIMAC_DIAG_EPI
__device__ unsigned long long crimac_epi_buf[1024 * 8];
#define CRIMAC_EPI_STAMP(k)                                                                                   \
  { unsigned long long tn_; __builtin_amdgcn_sched_barrier(0);                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (threadIdx.x == 0) crimac_epi_buf[((blockIdx.y * gridDim.x + blockIdx.x) % 1024) * 8 + (k)] = tn_; }
#else
#define CRIMAC_EPI_STAMP(k) {}
#endif

// Accumulator access for the two MFMA shapes (C/D layouts: cdna_hip_programming.md §3):
//   32x32x16: acc[MT][NT] of f32x16, row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane & 31
//   16x16x32: acc[MT][NT] of f32x4,  row = (lane>>4)*4 + r,               col = lane & 15
template <typename ACC> struct AccLayout;
template <> struct AccLayout<f32x16> {
  static constexpr int TS = 32, NR = 16;
  __device__ static __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};
template <> struct AccLayout<f32x4> {
  static constexpr int TS = 16, NR = 4;
  __device__ static __forceinline__ int row(int r, int lane) { return (lane >> 4) * 4 + r; }
  __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};

// acc[MT][NT]: wave (wr, wc) holds rows wr*MT*TS + i*TS + ..., cols wc*NT*TS + j*TS + ... .
//
// MODE (compile time: 0 none, 1 BatchNorm statistics, 2 BatchNorm-backward sums) and FULL (the tile lies inside the
// image: no per-element tests) are template parameters of the body -- the first form of this epilogue carried the
// three modes and the border tests through one runtime path: 2 800 vector instructions, 155 spilled SGPRs, ~19 k
// cycles per workgroup of the channel-split kernel (two 64-channel chunks of MFMAs take 18 k), phase stamps of
// tools/diag_wch_phases.py.  Stores (and the mode-2 loads) are buffer operations on a resource rebased at the tile's
// first pixel: 32-bit offsets, `row * pitch` instead of 64-bit pixel arithmetic per row, out of range = dropped.
// TA: storage type of the OUTPUT (bf16_t / half_t / float, or hp_t: fp16 plane pairs, staged as fp32 and split when
// the tile is stored).  PASSES > 1: the tile goes through the staging area in PASSES slices of BM / PASSES rows (fp32
// staging of a 256 x 128 tile would need 135 KB of LDS: one workgroup per CU instead of two).
// ILV: M tile i of wave row group wr (two groups) is tile row  wr * MT/2 + i  (i < MT/2)  or  MT + wr * MT/2 + i - MT/2:
// every slice of MT tile rows then holds half of each wave's accumulators (conv3x3_glds.hip, tall form).
template <typename TA, int BN, int BM, int NTHREADS, int MT, int NT, int MODE, bool FULL, typename ACC, int PASSES = 1,
          bool ILV = false>
__device__ __forceinline__ void conv_epilogue_body(const ACC (&acc)[MT][NT], const EpiParams& e,
                                                   unsigned char* smem, int b, int y0, int x0, int n0,
                                                   int wr, int wc, const float* bias_pre = nullptr) {
  CRIMAC_EPI_STAMP(0)
  using L = AccLayout<ACC>;
  constexpr bool HPO = __is_same(TA, hp_t);
  using TS = typename std::conditional<HPO, float, TA>::type;           // element type of the staging tile
  static_assert(!HPO || MODE != 2, "a plane-pair output is an MFMA operand, never the `da` of a BatchNorm block");
  constexpr int STAGE_PITCH = BN * (int)sizeof(TS) + 16;
  constexpr bool F32 = sizeof(TS) == 4;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RPASS = BM / PASSES;                                     // rows per pass
  static_assert(BM % PASSES == 0 && RPASS % 32 == 0, "a pass covers whole pairs of image rows");
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned char* stage = smem;                                           // [RPASS][STAGE_PITCH]
  float* sstat = reinterpret_cast<float*>(smem + RPASS * STAGE_PITCH);  // [2][BN]
  constexpr int mode = MODE;
  if (mode)
    for (int i = tid; i < 2 * BN; i += NTHREADS) sstat[i] = 0.f;
  float cs1[NT], cs2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; }
  float d1[8], d2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { d1[k] = 0.f; d2[k] = 0.f; }
  // (mode 2) BatchNorm constants of this thread's 8-channel chunk c8 = tid % (BN / 8): loaded once, as the bias below
  float sc[8], sh[8], mu[8], is[8];
  if (mode == 2) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = n0 + (tid % (BN / 8)) * 8 + k;
      mu[k] = e.bnb_vec[c];
      is[k] = e.bnb_vec[e.bnb_stride + c];
      sc[k] = e.bnb_vec[2 * e.bnb_stride + c];
      sh[k] = e.bnb_vec[3 * e.bnb_stride + c];
    }
  }
  // the bias is loaded ONCE, in front of the first slice: a load inside the slice loop would sit behind the previous
  // slice's global stores in the (in-order) vector-memory counter -- its wait is a wait for their acknowledgement
#ifdef CRIMAC_EPI_NO_HOIST              // (A/B builds: the bias load inside the slice loop, as before)
  constexpr bool HOIST = false;
#else
  constexpr bool HOIST = PASSES > 1;
#endif
  // bias_pre: the caller loaded this lane's NT bias values before its main loop (the load's L2 round trip at the head of
  // the epilogue is ~1.5 k cycles in which the whole workgroup does nothing)
  float bvs[NT];
  if (bias_pre) {
#pragma unroll
    for (int j = 0; j < NT; ++j) bvs[j] = bias_pre[j];
  } else if constexpr (HOIST) {
#pragma unroll
    for (int j = 0; j < NT; ++j) bvs[j] = e.bias ? e.bias[n0 + wc * (NT * L::TS) + j * L::TS + L::col(lane)] : 0.f;
  }
#pragma unroll
 for (int ps = 0; ps < PASSES; ++ps) {
  if (ps > 0) epi_barrier_lds();                                         // the previous slice has been read out (its stores may be in flight)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = wc * (NT * L::TS) + j * L::TS + L::col(lane);
    float bv;
    if (bias_pre) {
      bv = bvs[j];
      asm volatile("" : "+v"(bv));
    } else if constexpr (HOIST) {
      bv = bvs[j];
      asm volatile("" : "+v"(bv));     // (keeps this slice's bias / ReLU / rounding arithmetic behind the barrier above: hoisted
                                       // in front of it, hipcc holds both slices' results and spills ~80 registers)
    } else {
      bv = e.bias ? e.bias[n0 + col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row0 = ILV ? ((i < MT / 2 ? wr * (MT / 2) + i : MT + wr * (MT / 2) + i - MT / 2) * L::TS)
                           : wr * (MT * L::TS) + i * L::TS;              // (wave-uniform: an M tile lies in one pass)
      if (PASSES > 1 && row0 / RPASS != ps) continue;
#pragma unroll
      for (int r = 0; r < L::NR; ++r) {
        const int row = row0 + L::row(r, lane);
        float v = acc[i][j][r] * e.acc_scale + bv;
        if (e.relu) v = fmaxf(v, 0.f);
        const TS q = (TS)v;
        *reinterpret_cast<TS*>(stage + (row - ps * RPASS) * STAGE_PITCH + col * (int)sizeof(TS)) = q;
        if (mode == 1) {
          const float vs = e.stat_raw ? v : (HPO ? storage_round<hp_t>(v) : (float)q);      // statistics of the value as STORED
          const bool ok = FULL || ((y0 + (row >> 4) < e.H) && (x0 + (row & 15) < e.W));
          cs1[j] += ok ? vs : 0.f;
          cs2[j] += ok ? vs * vs : 0.f;
        }
      }
    }
  }
  if (ps == 0) { CRIMAC_EPI_STAMP(1) }        // slice 0 staged
  epi_barrier_lds();
  if (ps == 0) { CRIMAC_EPI_STAMP(2) }        // ... and visible
  if (mode == 1 && ps == PASSES - 1) {
    // rows live in registers and in the lane groups above the column lanes: fold with shuffles,
    // then one LDS add per column
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float t1 = cs1[j], t2 = cs2[j];
#pragma unroll
      for (int o = L::TS; o < 64; o <<= 1) {
        t1 += __shfl_xor(t1, o, 64);
        t2 += __shfl_xor(t2, o, 64);
      }
      if (lane < L::TS) {
        const int col = wc * (NT * L::TS) + j * L::TS + lane;
        atomicAdd(&sstat[col], t1);
        atomicAdd(&sstat[BN + col], t2);
      }
    }
  }
  if constexpr (F32 && MODE != 2) {
    // 4-byte outputs without the fused BatchNorm-backward sums: one 16-byte store per lane and CONSECUTIVE lanes on
    // consecutive 16 bytes, so that a wave-instruction writes whole 128-byte lines.  (The 8-channels-per-thread form
    // below writes a thread's 32 bytes as two instructions of 16 bytes at a 32-byte lane stride: every instruction
    // touches twice the lines and fills each only half.)  fp32: a lane takes 4 channels; plane pairs: a PAIR of lanes takes
    // an 8-channel group, the even lane stores its hi plane, the odd lane its lo plane.
    constexpr int CPR4 = BN / 4;                // 16-byte pieces per row
    constexpr int RPP4 = NTHREADS / CPR4;       // rows per store round
    const int c4 = tid % CPR4, r0 = tid / CPR4;
    TA* outp = reinterpret_cast<TA*>(e.out);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        outp + (((long)b * e.H + y0) * e.W + x0) * e.out_ld + n0, 0, 0x7FFFFFFF, 0x00020000);
    constexpr int UNR4 = HPO ? 4 : RPASS / RPP4;     // (plane pairs: 8 values + two planes live per row)
#pragma unroll(UNR4)
    for (int rr = 0; rr < RPASS / RPP4; ++rr) {
      const int lrow = r0 + rr * RPP4, row = ps * RPASS + lrow;
      const int ty = row >> 4, tx = row & 15;
      const bool ok = FULL || ((y0 + ty < e.H) && (x0 + tx < e.W));
      const int off = (int)(ok ? (unsigned)((((long)ty * e.W + tx) * e.out_ld + c4 * 4) * 4) : OOB);
      if constexpr (HPO) {
        float v[8];
        load8(reinterpret_cast<const float*>(stage + lrow * STAGE_PITCH) + (c4 >> 1) * 8, v);
        u32x4 hi, lo;
        hp_split(v, hi, lo);
        __builtin_amdgcn_raw_buffer_store_b128((c4 & 1) ? lo : hi, ro, off, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(
            *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(stage + lrow * STAGE_PITCH) + c4 * 4), ro, off, 0, 0);
      }
    }
  } else {
    constexpr int CPR = BN / 8;                 // 8-channel chunks per row
    constexpr int RPP = NTHREADS / CPR;         // rows per store round
    static_assert(RPP % 16 == 0 || 16 % RPP == 0, "a round covers whole image rows or a fraction of one");
    const int c8 = tid % CPR, r0 = tid / CPR;
    TA* outp = reinterpret_cast<TA*>(e.out);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        outp + (((long)b * e.H + y0) * e.W + x0) * e.out_ld + n0, 0, 0x7FFFFFFF, 0x00020000);
    __amdgpu_buffer_rsrc_t ry = ro;
    if (mode == 2)
      ry = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<TA*>(reinterpret_cast<const TA*>(e.bnb_y) + (((long)b * e.H + y0) * e.W + x0) * e.bnb_y_ld + n0), 0,
          0x7FFFFFFF, 0x00020000);
    // this thread's rows: ps * RPASS + r0 + rr * RPP -> (image row, column) of the tile; byte offsets from the tile's first pixel
    const int ty0 = (ps * RPASS + r0) >> 4, tx = r0 & 15;
    const unsigned o0 = (unsigned)((((long)ty0 * e.W + tx) * e.out_ld + c8 * 8) * (int)sizeof(TA));
    const unsigned o_step = (unsigned)(((long)(RPP >> 4) * e.W * e.out_ld + (RPP & 15) * e.out_ld) * (int)sizeof(TA));
    const unsigned q0 = mode == 2 ? (unsigned)((((long)ty0 * e.W + tx) * e.bnb_y_ld + c8 * 8) * (int)sizeof(TA)) : 0u;
    const unsigned q_step = mode == 2 ? (unsigned)(((long)(RPP >> 4) * e.W * e.bnb_y_ld + (RPP & 15) * e.bnb_y_ld) * (int)sizeof(TA)) : 0u;
    static_assert(RPP >= 16, "rows of one thread differ by whole image rows");
#ifdef CRIMAC_EPI_BNB_INLINE            // (A/B builds: the y load of a row right behind that row's store, as before)
    constexpr bool YFIRST = false;
#else
    constexpr bool YFIRST = MODE == 2;
#endif
    if constexpr (YFIRST) {
      // Mode 2, the saved forward outputs FIRST: all y loads of the slice (YB rows in flight) and the sums, then the
      // stores.  Vector-memory operations retire in issue order: a y load issued behind a row's store is a wait for that
      // store's acknowledgement -- 8-16 serialised round trips per slice (128 -> 128 @128x128 plane pairs: the input
      // gradient with the fused sums took 351 us, the same convolution with statistics 288).
      constexpr int NR = RPASS / RPP, YB = NR < 4 ? NR : 4, YV = F32 ? 2 : 1;
#pragma unroll 1
      for (int rb = 0; rb < NR; rb += YB) {
        u32x4 t[YB][YV];
        bool okb[YB];
#pragma unroll
        for (int u = 0; u < YB; ++u) {
          const int rr = rb + u, row = ps * RPASS + r0 + rr * RPP;
          okb[u] = FULL || ((y0 + (row >> 4) < e.H) && (x0 + tx < e.W));
          const int yoff = (int)(okb[u] ? q0 + rr * q_step : OOB);
#pragma unroll
          for (int h = 0; h < YV; ++h) t[u][h] = __builtin_amdgcn_raw_buffer_load_b128(ry, yoff, 16 * h, 0);
        }
#pragma unroll
        for (int u = 0; u < YB; ++u) {
          const TS* sp = reinterpret_cast<const TS*>(stage + (r0 + (rb + u) * RPP) * STAGE_PITCH) + c8 * 8;
          float g[8], yv[8];
          load8(sp, g);
          load8(reinterpret_cast<const TS*>(t[u]), yv);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float dz = (okb[u] && (yv[k] * sc[k] + sh[k]) > 0.f) ? g[k] : 0.f;
            d1[k] += dz;
            d2[k] += dz * (yv[k] - mu[k]) * is[k];
          }
        }
      }
    }
    // (sliced tiles with the fused BatchNorm-backward sums: the registers of y and g per row stand next to the accumulators
    // of the slices still to come -- fully unrolled, hipcc hoists every load and spills 70-100 registers)
    constexpr int UNR = (MODE == 2 && PASSES > 1) ? 2 : RPASS / RPP;
#pragma unroll(UNR)
    for (int rr = 0; rr < RPASS / RPP; ++rr) {
      const int lrow = r0 + rr * RPP, row = ps * RPASS + lrow;
      const bool ok = FULL || ((y0 + (row >> 4) < e.H) && (x0 + tx < e.W));
      const TS* sp = reinterpret_cast<const TS*>(stage + lrow * STAGE_PITCH) + c8 * 8;
      const int off = (int)(ok ? o0 + rr * o_step : OOB);
      if constexpr (HPO) {
        float v[8];
        load8(sp, v);
        u32x4 hi, lo;
        hp_split(v, hi, lo);
        __builtin_amdgcn_raw_buffer_store_b128(hi, ro, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(lo, ro, off, 16, 0);
      } else if constexpr (F32) {
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sp), ro, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sp + 4), ro, off, 16, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sp), ro, off, 0, 0);
      }
      if (mode == 2 && !YFIRST) {
        float g[8], yv[8];
        load8(sp, g);
        const int yoff = (int)(ok ? q0 + rr * q_step : OOB);
        if constexpr (F32) {
          u32x4 t[2];
          t[0] = __builtin_amdgcn_raw_buffer_load_b128(ry, yoff, 0, 0);
          t[1] = __builtin_amdgcn_raw_buffer_load_b128(ry, yoff, 16, 0);
          load8(reinterpret_cast<const TS*>(t), yv);
        } else {
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ry, yoff, 0, 0);
          load8(reinterpret_cast<const TS*>(&t), yv);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dz = (ok && (yv[k] * sc[k] + sh[k]) > 0.f) ? g[k] : 0.f;
          d1[k] += dz;
          d2[k] += dz * (yv[k] - mu[k]) * is[k];
        }
      }
    }
    if (mode == 2 && ps == PASSES - 1) {
      // threads of one chunk sit CPR lanes apart: fold inside the wave, then one LDS add per wave
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) {
          d1[k] += __shfl_xor(d1[k], o, 64);
          d2[k] += __shfl_xor(d2[k], o, 64);
        }
      }
      if (lane < CPR) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          atomicAdd(&sstat[c8 * 8 + k], d1[k]);
          atomicAdd(&sstat[BN + c8 * 8 + k], d2[k]);
        }
      }
    }
  }
  if (ps == 0) { CRIMAC_EPI_STAMP(3) }        // slice 0's stores issued
  if (e.pool_out) {
    // the staged slice still holds the values as stored (plane pairs: the fp32 values, rounding is monotone so the
    // maximum of the stored values is the stored maximum): 2x2 windows never straddle tiles or slices (y0, x0 even)
    constexpr int CPR = BN / 8;
    constexpr int PROWS = RPASS / 32;                   // pooled rows of the slice (rows / 2), 8 pooled columns
    TA* pool = reinterpret_cast<TA*>(e.pool_out);
    const int Hp = e.H >> 1, Wp = e.W >> 1;
    for (int i = tid; i < PROWS * 8 * CPR; i += NTHREADS) {
      const int c8 = i % CPR, pp = i / CPR;
      const int pyl = pp >> 3, pxl = pp & 7;
      const int yp = (y0 >> 1) + ps * PROWS + pyl, xp = (x0 >> 1) + pxl;
      if (yp >= Hp || xp >= Wp) continue;
      const int r00 = (2 * pyl) * 16 + 2 * pxl;
      float m[8], v[8];
      load8(reinterpret_cast<const TS*>(stage + r00 * STAGE_PITCH) + c8 * 8, m);
#pragma unroll
      for (int d = 1; d < 4; ++d) {
        load8(reinterpret_cast<const TS*>(stage + (r00 + (d >> 1) * 16 + (d & 1)) * STAGE_PITCH) + c8 * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
      }
      store8(pool + (((long)b * Hp + yp) * Wp + xp) * e.pool_ld + n0 + c8 * 8, m);
    }
  }
 }   // passes
  CRIMAC_EPI_STAMP(4)                         // all slices stored
  if (mode) {
    epi_barrier_lds();
    // thousands of workgroups add to the same N channels: spread them over replicas (the atomic
    // unit serialises same-address adds); the consumer sums the replicas
    const long rep = (long)(blockIdx.x % (unsigned)e.stat_replicas) * e.N;
    for (int c = tid; c < BN; c += NTHREADS) {
      atomicAdd(&e.stat_sum[rep + n0 + c], (double)sstat[c]);
      atomicAdd(&e.stat_sumsq[rep + n0 + c], (double)sstat[BN + c]);
    }
  }
  CRIMAC_EPI_STAMP(5)
}

// Dispatch on the (workgroup-uniform) tile position and, when MODE < 0, on the runtime statistics mode.
template <typename TA, int BN, int BM, int NTHREADS, int MT, int NT, typename ACC, int MODE = -1, int PASSES = 1,
          bool ILV = false>
__device__ __forceinline__ void conv_epilogue(const ACC (&acc)[MT][NT], const EpiParams& e,
                                              unsigned char* smem, int b, int y0, int x0, int n0,
                                              int tile_rows, int wr, int wc, const float* bias_pre = nullptr) {
  const bool full = (y0 + tile_rows <= e.H) && (x0 + 16 <= e.W);
  const int mode = MODE >= 0 ? MODE : (e.stat_sum ? e.stat_mode : 0);
#define CRIMAC_EPI(M)                                                                                      \
  do {                                                                                                     \
    if (full) conv_epilogue_body<TA, BN, BM, NTHREADS, MT, NT, M, true, ACC, PASSES, ILV>(acc, e, smem, b, y0, x0, n0, wr, wc, bias_pre);   \
    else conv_epilogue_body<TA, BN, BM, NTHREADS, MT, NT, M, false, ACC, PASSES, ILV>(acc, e, smem, b, y0, x0, n0, wr, wc, bias_pre);       \
  } while (0)
  if constexpr (MODE >= 0) {
    CRIMAC_EPI(MODE);
  } else {
    if (mode == 0) CRIMAC_EPI(0);
    else if (mode == 1) CRIMAC_EPI(1);
    else if constexpr (!__is_same(TA, hp_t)) CRIMAC_EPI(2);      // (plane-pair outputs: rejected by the entry point)
  }
#undef CRIMAC_EPI
}
