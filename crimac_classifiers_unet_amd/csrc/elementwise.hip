// HBM-bound kernels of the U-Net hot path for CDNA4 (gfx950): BatchNorm statistics / apply /
// backward, ReLU, 2x2 max-pool and its backward, layout conversion, weight packing, SGD.
// All tensor accesses are 16-byte vectors over the contiguous NHWC channel axis; per-channel
// reductions keep partial sums in registers, combine them in LDS and issue one global atomic per
// channel per workgroup.
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

// ---- error text -------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void crimac_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* crimac_last_error(void) { return g_err; }
extern "C" int crimac_version(void) { return CRIMAC_ABI_VERSION; }
extern "C" int crimac_layer_desc_size(void) { return (int)sizeof(crimac_layer_desc); }

namespace {

// The streaming BatchNorm passes (bn_act, bn_act_pool, bn_bwd_apply) walk their rows from the END of the tensor: their input
// was written front to back by the convolution in front of them, so its last rows are what the memory-side cache still holds,
// and the rows they write last are the first ones the next convolution asks for.  Serialized bf16 step: bn_bwd_apply 1041 ->
// 1029 us, bn_act 505 -> 497, bn_act_pool 220 -> 213; timed step -0.03 to -0.06 ms in seven of seven alternating pairs on three
// boxes (tools/runs/r5_31.sh, r5_32.sh).  Nothing for the pool-backward passes (their inputs are older), which walk forward.
// -DCRIMAC_STREAM_FORWARD restores the forward walk for A/B builds.
#ifdef CRIMAC_STREAM_FORWARD
#define CRIMAC_ROW(x, M) (x)
#else
#define CRIMAC_ROW(x, M) ((M) - 1 - (x))
#endif

constexpr int kMaxBlocks = 2048;

inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// WHOLE ROUNDS for the grid-stride streaming kernels.  A grid that is not a whole number of resident rounds ends with a partly
// filled one in which the chip streams at a fraction of its rate: the 2048 workgroups of bn_bwd_apply at its three workgroups
// per CU are 2.67 rounds -- 5.6 TB/s on the level-0 tensors; 768 (one round) or 1536: 6.4-6.5 TB/s, 1024: 5.4
// (profiles/r05_elementwise_whole_rounds.txt).  So: never more workgroups than are resident at once
// (CUs x hipOccupancyMaxActiveBlocksPerMultiprocessor of that kernel; fewer workgroups also means fewer prologues, which in the
// replica-summing kernels are not free).  Looked up once per (kernel, device).
struct ResidentEntry { const void* fn; int dev, blocks; };
static ResidentEntry g_resident[96];
static int g_resident_n = 0;
static std::mutex g_resident_mu;
template <typename K>
inline int whole_rounds(K kernel, size_t lds, int grid, int block = 256) {
  static const int off = getenv("CRIMAC_WHOLE_ROUNDS") ? atoi(getenv("CRIMAC_WHOLE_ROUNDS")) == 0 : 0;      // (A/B runs)
  int dev = 0;
  if (off || grid <= 256 || hipGetDevice(&dev) != hipSuccess) return grid;
  const void* fn = reinterpret_cast<const void*>(kernel);
  int blocks = 0;
  {
    std::lock_guard<std::mutex> lk(g_resident_mu);
    for (int i = 0; i < g_resident_n; ++i)
      if (g_resident[i].fn == fn && g_resident[i].dev == dev) blocks = g_resident[i].blocks;
    if (blocks == 0) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) == hipSuccess &&
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
        blocks = per_cu * cus;
      else
        blocks = -1;
      if (g_resident_n < 96) g_resident[g_resident_n++] = ResidentEntry{fn, dev, blocks};
    }
  }
  return (blocks > 0 && grid > blocks) ? blocks : grid;
}

// pixel index -> (image, pixel in image); 32-bit divide when the operands fit (a 64-bit divide is ~4x the
// instructions, noticeable in the one-pixel-per-thread kernels)
__device__ __forceinline__ void pix_split(long p, long HW, long& b, long& hw) {
  if (((p | HW) >> 31) == 0) {
    const unsigned q = (unsigned)p / (unsigned)HW;
    b = q;
    hw = (unsigned)p - q * (unsigned)HW;
  } else {
    b = p / HW;
    hw = p % HW;
  }
}

// ---- layout: NCHW fp32 -> NHWC (channel padded) -----------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int C, long HW,
                                    long npix, long ld) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix;
       p += (long)gridDim.x * blockDim.x) {
    long b, hw;
    pix_split(p, HW, b, hw);
    const float* src = in + b * C * HW + hw;
    T* dst = out + p * ld;
    for (int c0 = 0; c0 < ld; c0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c0 + j) < C ? src[(long)(c0 + j) * HW] : 0.f;
      store8(dst + c0, v);
    }
  }
}

// ---- weight packing -----------------------------------------------------------------------------
// value -> plane 0 in `hi`, planes 1..npl-1 in `lo` (plane stride `n`)
__device__ __forceinline__ void put_planes(float v, int npl, int fp16, unsigned short* hi, unsigned short* lo,
                                           long i, long n) {
  unsigned short b = f2bits16(v, fp16);
  hi[i] = b;
  for (int k = 1; k < npl; ++k) {
    v -= bits162f(b, fp16);
    b = f2bits16(v, fp16);
    lo[(long)(k - 1) * n + i] = b;
  }
}

// the same for interleaved plane pairs: element `col` of row `row` (rows of `rowlen` channels)
__device__ __forceinline__ void put_planes_il(float v, int fp16, unsigned short* buf, long row, int col, int rowlen) {
  unsigned short* dst = buf + row * 2L * rowlen + il_pos(col, rowlen);
  const unsigned short b = f2bits16(v, fp16);
  dst[0] = b;
  dst[il_cb(rowlen)] = f2bits16(v - bits162f(b, fp16), fp16);
}

__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int Co, int Ci, int Ci_pad,
                                    const float* __restrict__ scale, int npl, unsigned short* fwd_hi,
                                    unsigned short* fwd_lo, unsigned short* dg_hi,
                                    unsigned short* dg_lo) {
  const PlaneFmt pf = plane_fmt(npl);
  const long n_fwd = 9L * Co * Ci_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_fwd;
       i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci_pad);
    const long r = i / Ci_pad;
    const int co = (int)(r % Co), t = (int)(r / Co);
    float v = 0.f;
    if (ci < Ci) {
      v = w[((long)co * Ci + ci) * 9 + t];
      if (scale) v *= scale[co];
    }
    if ((npl & CRIMAC_PLANES_FWD_FRAG) && pf.interleaved) {      // fragment-major (CRIMAC_EPI_WFRAG) on the row of 2 Ci_pad halves
      const int c = (int)il_pos(ci, Ci_pad);
      const unsigned short b = f2bits16(v * pf.fwd_scale, pf.fwd_fp16);
      fwd_hi[wfrag_index(t, co, c, Co, 2 * Ci_pad)] = b;
      fwd_hi[wfrag_index(t, co, c + il_cb(Ci_pad), Co, 2 * Ci_pad)] = f2bits16(v * pf.fwd_scale - bits162f(b, pf.fwd_fp16), pf.fwd_fp16);
    } else if (npl & CRIMAC_PLANES_FWD_FRAG) {
      put_planes(v * pf.fwd_scale, pf.npl, pf.fwd_fp16, fwd_hi, fwd_lo, wfrag_index(t, co, ci, Co, Ci_pad), n_fwd);
    } else if (pf.interleaved) put_planes_il(v * pf.fwd_scale, pf.fwd_fp16, fwd_hi, r, ci, Ci_pad);
    else put_planes(v * pf.fwd_scale, pf.npl, pf.fwd_fp16, fwd_hi, fwd_lo, i, n_fwd);
  }
  if (dg_hi) {
    const long n_dg = 9L * Ci * Co;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_dg;
         i += (long)gridDim.x * blockDim.x) {
      const int co = (int)(i % Co);
      const long r = i / Co;
      const int ci = (int)(r % Ci), t = (int)(r / Ci);
      const float v = w[((long)co * Ci + ci) * 9 + (8 - t)] * pf.dg_scale;
      if (pf.interleaved) put_planes_il(v, pf.dg_fp16, dg_hi, r, co, Co);
      else put_planes(v, pf.npl, pf.dg_fp16, dg_hi, dg_lo, i, n_dg);
    }
  }
}

__global__ void pack_upconv_kernel(const float* __restrict__ w, int Ci, int Co, int npl,
                                   unsigned short* fwd_hi, unsigned short* fwd_lo,
                                   unsigned short* dg_hi, unsigned short* dg_lo) {
  // w[ci][co][a][b]; fwd[(ab*Co + co)][ci]; dgrad[ab][ci][co]
  const PlaneFmt pf = plane_fmt(npl);
  const long n = 4L * Ci * Co;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    {
      const int ci = (int)(i % Ci);
      const long r = i / Ci;
      const int co = (int)(r % Co), ab = (int)(r / Co);
      const float v = w[((long)ci * Co + co) * 4 + ab] * pf.fwd_scale;
      if (pf.interleaved) put_planes_il(v, pf.fwd_fp16, fwd_hi, r, ci, Ci);
      else put_planes(v, pf.npl, pf.fwd_fp16, fwd_hi, fwd_lo, i, n);
    }
    if (dg_hi) {
      const int co = (int)(i % Co);
      const long r = i / Co;
      const int ci = (int)(r % Ci), ab = (int)(r / Ci);
      const float v = w[((long)ci * Co + co) * 4 + ab] * pf.dg_scale;
      if (pf.interleaved) put_planes_il(v, pf.dg_fp16, dg_hi, r, co, Co);
      else put_planes(v, pf.npl, pf.dg_fp16, dg_hi, dg_lo, i, n);
    }
  }
}

__global__ void unpack_wgrad_conv_kernel(const float* __restrict__ dw, int Co, int Ci, int Ci_pad,
                                         float* __restrict__ grad) {
  const long n = (long)Co * Ci * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % 9);
    const long r = i / 9;
    const int ci = (int)(r % Ci), co = (int)(r / Ci);
    grad[i] = dw[((long)t * Co + co) * Ci_pad + ci];
  }
}

__global__ void unpack_wgrad_upconv_kernel(const float* __restrict__ dw, int Ci, int Co,
                                           float* __restrict__ grad) {
  const long n = (long)Ci * Co * 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const int ab = (int)(i % 4);
    const long r = i / 4;
    const int co = (int)(r % Co), ci = (int)(r / Co);
    grad[i] = dw[((long)ab * Ci + ci) * Co + co];
  }
}

// ---- per-channel reductions over an [M][C] (pixel-major) matrix ---------------------------------
// Thread -> (8-channel chunk, row lane).  OP produces up to two quantities per element.
template <typename T, typename ACC, int NQ, int UNR = 2, typename OP>
__device__ __forceinline__ void colreduce(long M, int C, OP op, ACC* out0, ACC* out1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);   // [NQ][C]
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;     // rows per iteration
  for (int i = threadIdx.x; i < NQ * C; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (rl < rpi && cpr <= 256) {
    const long stride = (long)gridDim.x * rpi;
    long m = (long)blockIdx.x * rpi + rl;
    for (; m + (UNR - 1) * stride < M; m += UNR * stride) {      // UNR independent rows in flight
      float q0[UNR][8], q1[UNR][8];
#pragma unroll
      for (int u = 0; u < UNR; ++u) op(m + u * stride, chunk * 8, q0[u], q1[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s0[j] += q0[u][j];
          if (NQ > 1) s1[j] += q1[u][j];
        }
    }
    for (; m < M; m += stride) {
      float q0[8], q1[8];
      op(m, chunk * 8, q0, q1);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += q0[j]; if (NQ > 1) s1[j] += q1[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&lds[chunk * 8 + j], s0[j]);
      if (NQ > 1) atomicAdd(&lds[C + chunk * 8 + j], s1[j]);
    }
  }
  __syncthreads();
  if (!out0) return;        // nothing to accumulate into (bn_bwd_apply without a bias gradient)
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(&out0[c], (ACC)lds[c]);
    if (NQ > 1) atomicAdd(&out1[c], (ACC)lds[C + c]);
  }
}

template <typename T, typename ACC, int NQ>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ y, long ld, long M, int C,
                                                       ACC* sum, ACC* sumsq) {
  colreduce<T, ACC, NQ>(M, C,
                        [&](long m, int c0, float (&q0)[8], float (&q1)[8]) {
                          load8(y + m * ld + c0, q0);
#pragma unroll
                          for (int j = 0; j < 8; ++j) q1[j] = q0[j] * q0[j];
                        },
                        sum, sumsq);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const double* sum, const double* sumsq, int nrep, long M, int C, const float* gamma,
    const float* beta, float eps, float momentum, float* rmean, float* rvar, long long* nbt,
    float* mean_o, float* invstd_o, float* scale_o, float* shift_o) {
  // 16 channels per workgroup, 16 threads per channel striding over the replicas
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int r = rl; r < nrep; r += 16) { s1 += sum[(long)r * C + c]; s2 += sumsq[(long)r * C + c]; }
  red[0][cl][rl] = s1;
  red[1][cl][rl] = s2;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  s1 = 0.0; s2 = 0.0;
  for (int r = 0; r < 16; ++r) { s1 += red[0][cl][r]; s2 += red[1][cl][r]; }
  const double mean = s1 / (double)M;
  double var = s2 / (double)M - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  mean_o[c] = (float)mean;
  invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = beta[c] - (float)mean * sc;
  if (rmean) {
    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// dst[i] = sum over replicas of src[r*stride + i]  (fp64 and/or fp32 destination);
// 16 channels per workgroup, 16 threads per channel striding over the replicas; blockIdx.y = 1 does
// the same for the second (src_b -> dst_b) pair
__global__ __launch_bounds__(256) void sum_replicas_kernel(const double* __restrict__ src, int nrep,
                                                           long stride, int n, double* dst64, float* dst32,
                                                           const double* __restrict__ src_b, double* dst_b) {
  __shared__ double red[16][17];
  if (blockIdx.y == 1) { src = src_b; dst64 = dst_b; dst32 = nullptr; }
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (i < n)
    for (int r = rl; r < nrep; r += 16) s += src[(long)r * stride + i];
  red[cl][rl] = s;
  __syncthreads();
  if (rl != 0 || i >= n) return;
  s = 0.0;
  for (int r = 0; r < 16; ++r) s += red[cl][r];
  if (dst64) dst64[i] = s;
  if (dst32) dst32[i] = (float)s;
}

// ---- training-mode BatchNorm statistics finished INSIDE the consumer -------------------------------------------
// (no bn_finalize / sum_replicas launch between the convolution and the kernel that applies the statistics: 36 launches
// of a step and their dependency bubbles.)  Every workgroup sums the producer's `nrep` replica accumulators of all C
// channels itself -- C x nrep x 16 bytes out of L2, the engine keeps C x nrep <= 4096 -- and holds the per-channel
// constants in LDS; workgroup 0 also writes them out (the backward pass reads them) and updates the running statistics.
// All workgroups add in the same order: every one of them sees the same constants.
struct BnFin {
  const double* sum; const double* sumsq;   // [nrep][C]
  int nrep;
  long count;                               // pixels the sums were taken over
  const float* gamma; const float* beta;
  float eps, momentum;
  float* rmean; float* rvar; long long* nbt;   // running statistics (may be null)
  float* vec; long stride;                  // rows mean | invstd | scale | shift, row stride `stride`
};
// sums of two replica arrays for all C channels; every thread of the 256-thread block calls it; `red`: 512 doubles of
// LDS.  fn(c, s1, s2) runs once per channel (by one thread of the block).
template <typename F>
__device__ __forceinline__ void replica_sums_block(const double* __restrict__ a, const double* __restrict__ b, int nrep,
                                                   int C, double* red, F&& fn) {
  const int tid = threadIdx.x;
  const int k = C >= 256 ? 1 : 256 / C;          // threads per channel
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + (k == 1 ? tid : tid % C), j = k == 1 ? 0 : tid / C;
    double s1 = 0.0, s2 = 0.0;
    if (c < C && j < k) {
      // all of a batch's loads are issued before the first add: one memory latency per 8 replicas, not one per replica
      // (the accumulators were filled by memory-side atomics: these reads miss the L2)
      constexpr int NB = 8;
      int r = j;
      for (; r + (NB - 1) * k < nrep; r += NB * k) {
        double v1[NB], v2[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) { v1[u] = a[(long)(r + u * k) * C + c]; v2[u] = b[(long)(r + u * k) * C + c]; }
#pragma unroll
        for (int u = 0; u < NB; ++u) { s1 += v1[u]; s2 += v2[u]; }
      }
      double w1[NB], w2[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const bool in = r + u * k < nrep;
        w1[u] = in ? a[(long)(r + u * k) * C + c] : 0.0;
        w2[u] = in ? b[(long)(r + u * k) * C + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) { s1 += w1[u]; s2 += w2[u]; }
    }
    if (k > 1) {
      red[tid] = s1;
      red[256 + tid] = s2;
      __syncthreads();
      if (j == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int q = 0; q < k; ++q) { s1 += red[q * C + c]; s2 += red[256 + q * C + c]; }
      }
    }
    if (j == 0 && c < C) fn(c, s1, s2);
  }
}
// cst: [4][C] floats of LDS (mean | invstd | scale | shift), valid after the call (it ends with a barrier)
__device__ __forceinline__ void bn_fin_block(const BnFin& a, int C, float* cst, double* red) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.nbt) *a.nbt += 1;
  const bool writer = blockIdx.x == 0;
  replica_sums_block(a.sum, a.sumsq, a.nrep, C, red, [&](int c, double s1, double s2) {
    const double mean = s1 / (double)a.count;
    double var = s2 / (double)a.count - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float sc = a.gamma[c] * invstd;
    const float sh = a.beta[c] - (float)mean * sc;
    cst[c] = (float)mean; cst[C + c] = invstd; cst[2 * C + c] = sc; cst[3 * C + c] = sh;
    if (writer) {
      a.vec[c] = (float)mean; a.vec[a.stride + c] = invstd; a.vec[2 * a.stride + c] = sc; a.vec[3 * a.stride + c] = sh;
      if (a.rmean) {
        const double unb = a.count > 1 ? var * ((double)a.count / (double)(a.count - 1)) : var;
        a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * (float)mean;
        a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * (float)unb;
      }
    }
  });
  __syncthreads();
}

// ---- BN apply (+ReLU) (+2x2 max-pool) -------------------------------------------------------------
// T: storage type of the convolution output y; TO: of the activation (H3P: fp32 in, fp16 plane pairs out)
// FIN: scale / shift come from the producer's replica accumulators (bn_fin_block) instead of two finished vectors
template <typename T, typename TO = T, bool FIN = false>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ y, long y_ld,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu,
                                                     TO* __restrict__ out, long out_ld, long M, int C, BnFin fin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
  float* cst = reinterpret_cast<float*>(bn_smem + 512 * sizeof(double));
  if constexpr (FIN) bn_fin_block(fin, C, cst, reinterpret_cast<double*>(bn_smem));
  // thread -> fixed 8-channel chunk (its scale/shift live in registers), rows strided over the grid
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  if (rl >= rpi) return;
  const int c0 = chunk * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (FIN) {
      sc[j] = cst[2 * C + c0 + j];
      sh[j] = cst[3 * C + c0 + j];
    } else {
      sc[j] = scale ? scale[c0 + j] : 1.f;
      sh[j] = scale ? shift[c0 + j] : 0.f;
    }
  }
  const long stride = (long)gridDim.x * rpi;
  constexpr int U = 4;
  for (long m = (long)blockIdx.x * rpi + rl; m < M; m += U * stride) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (m + u * stride < M) load8s(y + CRIMAC_ROW(m + u * stride, M) * y_ld + c0, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (m + u * stride >= M) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[u][j] = v[u][j] * sc[j] + sh[j];
        if (relu) v[u][j] = fmaxf(v[u][j], 0.f);
      }
      store8(out + CRIMAC_ROW(m + u * stride, M) * out_ld + c0, v[u]);
    }
  }
}

template <typename T, typename TO = T, bool FIN = false>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const T* __restrict__ y, long y_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          TO* __restrict__ out, long out_ld,
                                                          TO* __restrict__ pool, long pool_ld, int B,
                                                          int H, int W, int C, BnFin fin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
  float* cst = reinterpret_cast<float*>(bn_smem + 512 * sizeof(double));
  if constexpr (FIN) bn_fin_block(fin, C, cst, reinterpret_cast<double*>(bn_smem));
  // thread -> fixed 8-channel chunk (its scale / shift live in registers), pooled pixels strided over the grid, the eight
  // vectors of TWO 2 x 2 windows in flight.  (The first form took one (window, chunk) item per thread and iteration: 64-bit
  // divisions and sixteen constant loads per item, four loads in flight -- 4.4 TB/s on the level-0 tensors where bn_act
  // reaches 6.3; tools/bench_stride.py.)
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  if (rl >= rpi) return;
  const int c0 = chunk * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (FIN) {
      sc[j] = cst[2 * C + c0 + j];
      sh[j] = cst[3 * C + c0 + j];
    } else {
      sc[j] = scale ? scale[c0 + j] : 1.f;
      sh[j] = scale ? shift[c0 + j] : 0.f;
    }
  }
  const int Hp = H / 2, Wp = W / 2;
  const long HWp = (long)Hp * Wp, npool = (long)B * HWp;
  const long stride = (long)gridDim.x * rpi;
  constexpr int U = 2;
  for (long m = (long)blockIdx.x * rpi + rl; m < npool; m += U * stride) {
    float v[U][4][8];
    long pix0[U], pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (m + u * stride >= npool) continue;
      long b, hw;
      pix_split(CRIMAC_ROW(m + u * stride, npool), HWp, b, hw);
      const int yp = (int)(hw / Wp), xp = (int)(hw - (long)yp * Wp);
      pix0[u] = (b * H + 2 * yp) * (long)W + 2 * xp;
      pp[u] = (b * Hp + yp) * (long)Wp + xp;
#pragma unroll
      for (int d = 0; d < 4; ++d) load8s(y + (pix0[u] + (d >> 1) * (long)W + (d & 1)) * y_ld + c0, v[u][d]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (m + u * stride >= npool) continue;
      float mx[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[u][d][j] = v[u][d][j] * sc[j] + sh[j];
          if (relu) v[u][d][j] = fmaxf(v[u][d][j], 0.f);
          mx[j] = d == 0 ? v[u][d][j] : fmaxf(mx[j], v[u][d][j]);
        }
        if (out) store8(out + (pix0[u] + (d >> 1) * (long)W + (d & 1)) * out_ld + c0, v[u][d]);
      }
      store8(pool + pp[u] * pool_ld + c0, mx);
    }
  }
}


// Fused BatchNorm+ReLU backward sums (== bn_bwd_reduce on the tensor a kernel has just produced): the
// producer keeps sum dz / sum dz*xhat of its thread-fixed 8-channel chunk in registers while it still
// holds `da`; at the end they go through LDS into one of `replicas` fp64 accumulators [replicas][C].
struct BnbArgs {
  const void* y;          // saved conv output of the BatchNorm layer `da` feeds, [pixels][y_ld]
  long y_ld;
  const float* vec;       // rows mean, invstd, scale, shift; row stride `stride`
  long stride;
  double* sum_dz;
  double* sum_dzx;
  int replicas;
};
struct BnbConst { float sc[8], sh[8], mu[8], is[8]; };
__device__ __forceinline__ void bnb_load(const BnbArgs& a, int c0, BnbConst& k) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    k.mu[j] = a.vec[c0 + j];
    k.is[j] = a.vec[a.stride + c0 + j];
    k.sc[j] = a.vec[2 * a.stride + c0 + j];
    k.sh[j] = a.vec[3 * a.stride + c0 + j];
  }
}
template <typename T>
__device__ __forceinline__ void bnb_accum(const BnbArgs& a, const BnbConst& k, long pix, int c0, const float (&g)[8],
                                          float (&d1)[8], float (&d2)[8]) {
  float yv[8];
  load8(reinterpret_cast<const T*>(a.y) + pix * a.y_ld + c0, yv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float dz = (yv[j] * k.sc[j] + k.sh[j]) > 0.f ? g[j] : 0.f;
    d1[j] += dz;
    d2[j] += dz * (yv[j] - k.mu[j]) * k.is[j];
  }
}
// same with y already in registers
__device__ __forceinline__ void bnb_accum_v(const BnbConst& k, const float (&yv)[8], const float (&g)[8],
                                            float (&d1)[8], float (&d2)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float dz = (yv[j] * k.sc[j] + k.sh[j]) > 0.f ? g[j] : 0.f;
    d1[j] += dz;
    d2[j] += dz * (yv[j] - k.mu[j]) * k.is[j];
  }
}
// lds: [2][C] floats, zeroed and synchronised by the caller; every thread of the block calls this
__device__ __forceinline__ void bnb_flush(const BnbArgs& a, float* lds, int C, int c0, const float (&d1)[8],
                                          const float (&d2)[8], int nthreads) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    atomicAdd(&lds[c0 + j], d1[j]);
    atomicAdd(&lds[C + c0 + j], d2[j]);
  }
  __syncthreads();
  const long rep = (long)(blockIdx.x % (unsigned)a.replicas) * C;
  for (int c = threadIdx.x; c < C; c += nthreads) {
    atomicAdd(&a.sum_dz[rep + c], (double)lds[c]);
    atomicAdd(&a.sum_dzx[rep + c], (double)lds[C + c]);
  }
}

// ---- backward of max-pool + skip add ---------------------------------------------------------------
// thread -> fixed 8-channel chunk, pooled pixels strided over the grid (channels contiguous across the lanes of
// a pixel); BNB: da feeds a BatchNorm+ReLU block whose backward sums are accumulated on the fly
// T: storage type of the gradients (and of y); TA: of the forward activation `a` (H3P: fp16 plane pairs)
template <typename T, bool BNB, typename TA = T>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((BNB && sizeof(T) == 2 && sizeof(TA) == 2) ? 3 : 1)))      // (16-bit + sums: <= 168 registers)
void unpool_add_kernel(const T* __restrict__ dp, long dp_ld,
                                                         const TA* __restrict__ a, long a_ld,
                                                         const T* __restrict__ ds, long ds_ld,
                                                         T* __restrict__ da, long da_ld, int B, int H,
                                                         int W, int C, BnbArgs bnb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);   // [2][C] (BNB only)
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  const int c0 = chunk * 8;
  const int Hp = H / 2, Wp = W / 2;
  const long npool = (long)B * Hp * Wp;
  BnbConst k;
  float d1[8], d2[8];
  if constexpr (BNB) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) lds[i] = 0.f;
    bnb_load(bnb, c0, k);
#pragma unroll
    for (int j = 0; j < 8; ++j) { d1[j] = 0.f; d2[j] = 0.f; }
    __syncthreads();
  }
  if (rl < rpi) {
    for (long r = (long)blockIdx.x * rpi + rl; r < npool; r += (long)gridDim.x * rpi) {
      const int xp = (int)(r % Wp);
      const long t = r / Wp;
      const int yp = (int)(t % Hp);
      const long b = t / Hp;
      if constexpr (BNB && sizeof(T) == 2 && sizeof(TA) == 2) {
        // 16-bit storage with the fused sums (round 5): the window's four y vectors stay PACKED (16 registers instead of
        // 32 + the 32 of the rebuilt activations) and are unpacked once for the arg-max and once for the sums: 174 -> <= 128
        // registers, i.e. four waves per SIMD instead of two under a kernel that was bound by its arithmetic at occupancy 2
        // (170 us at 256 x 256 against a 116 us byte floor).  Same arithmetic, same first-maximum rule: bit-identical.
        float g[8];
        load8s(dp + r * dp_ld + c0, g);
        long pixq[4];
        u16x8 yraw[4];
        float best[8];
        int arg[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          pixq[d] = (b * H + 2 * yp + (d >> 1)) * (long)W + 2 * xp + (d & 1);
#if CRIMAC_STREAM_NT
          yraw[d] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(reinterpret_cast<const T*>(bnb.y) + pixq[d] * bnb.y_ld + c0));
#else
          yraw[d] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const T*>(bnb.y) + pixq[d] * bnb.y_ld + c0);
#endif
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float yv[8];
          load8(reinterpret_cast<const T*>(&yraw[d]), yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a_ = storage_round<TA>(fmaxf(yv[j] * k.sc[j] + k.sh[j], 0.f));
            if (d == 0) { best[j] = a_; arg[j] = 0; }
            else if (a_ > best[j]) { best[j] = a_; arg[j] = d; }      // first maximum wins (aten max_pool2d)
          }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float o[8], yv[8];
          if (ds) load8s(ds + pixq[d] * ds_ld + c0, o);
          load8(reinterpret_cast<const T*>(&yraw[d]), yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = storage_round<T>((ds ? o[j] : 0.f) + (arg[j] == d ? g[j] : 0.f));
          if (da) store8(da + pixq[d] * da_ld + c0, o);
          bnb_accum_v(k, yv, o, d1, d2);
        }
        continue;
      }
      float g[8], av[4][8], yv[4][8];
      load8s(dp + r * dp_ld + c0, g);
      long pix[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        pix[d] = (b * H + 2 * yp + (d >> 1)) * (long)W + 2 * xp + (d & 1);
        if constexpr (BNB) {
          // the block's conv output y is read for the fused sums anyway: the pooled activation is rebuilt from it,
          // a = round(relu(y * scale + shift)) exactly as bn_act_pool stored it (same fma, same rounding, so the same
          // ties and the same first maximum) -- `a` is not read at all (one tensor less: 0.5 GB per step)
          load8s(reinterpret_cast<const T*>(bnb.y) + pix[d] * bnb.y_ld + c0, yv[d]);
#pragma unroll
          for (int j = 0; j < 8; ++j) av[d][j] = storage_round<TA>(fmaxf(yv[d][j] * k.sc[j] + k.sh[j], 0.f));
        } else {
          load8s(a + pix[d] * a_ld + c0, av[d]);
        }
      }
      int arg[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int best = 0;
        float bv = av[0][j];
#pragma unroll
        for (int d = 1; d < 4; ++d)
          if (av[d][j] > bv) { bv = av[d][j]; best = d; }   // first maximum wins (aten max_pool2d)
        arg[j] = best;
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float o[8];
        if (ds) load8s(ds + pix[d] * ds_ld + c0, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = (ds ? o[j] : 0.f) + (arg[j] == d ? g[j] : 0.f);
          if constexpr (BNB) o[j] = storage_round<T>(o[j]);          // the sums see da as it is stored
        }
        if (da) store8(da + pix[d] * da_ld + c0, o);      // (da == nullptr: sums only -- unpool_bn_bwd_apply rebuilds da)
        if constexpr (BNB) bnb_accum_v(k, yv[d], o, d1, d2);
      }
    }
  }
  if constexpr (BNB) bnb_flush(bnb, lds, C, c0, d1, d2, 256);
}

// dy of one element from (y, d(activation)) and the per-channel constants.  ONE spelling with explicit fused multiply-adds
// and contraction off, shared by every apply kernel: left to the compiler, the stand-alone and the unpool-fused kernel
// contracted the same source differently and 15 % of their fp32 results differed in the last bit.
__device__ __forceinline__ float bn_bwd_dy(float yv, float g, float sc, float sh, float mu, float is, float k1, float k2) {
#pragma clang fp contract(off)
  const float act = __builtin_fmaf(yv, sc, sh);
  const float dz = act > 0.f ? g : 0.f;
  const float xh = (yv - mu) * is;
  return sc * __builtin_fmaf(-xh, k2, dz - k1);
}

// ---- BatchNorm + ReLU backward ------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ da, long da_ld,
                                                            const T* __restrict__ y, long y_ld,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, long M,
                                                            int C, double* sum_dz, double* sum_dzx) {
  const int c0t = (threadIdx.x % (C / 8)) * 8;     // this thread's channel chunk (see colreduce)
  float sc[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[c0t + j]; sh[j] = shift[c0t + j]; mu[j] = mean[c0t + j]; is[j] = invstd[c0t + j];
  }
  colreduce<T, double, 2>(M, C,
                          [&](long m, int c0, float (&q0)[8], float (&q1)[8]) {
                            float g[8], yv[8];
                            load8(da + m * da_ld + c0, g);
                            load8(y + m * y_ld + c0, yv);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                              const float act = yv[j] * sc[j] + sh[j];
                              const float dz = act > 0.f ? g[j] : 0.f;
                              q0[j] = dz;
                              q1[j] = dz * (yv[j] - mu[j]) * is[j];
                            }
                          },
                          sum_dz, sum_dzx);
}

// T: storage type of da and y; TD: of the output gradient dy (H3P: fp16 plane pairs -- it feeds two contractions)
template <typename T, typename TD = T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ da, long da_ld, const T* __restrict__ y, long y_ld,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const double* __restrict__ sum_dz,
    const double* __restrict__ sum_dzx, long M, long count, int C, TD* __restrict__ dy, long dy_ld, float* dgamma,
    float* dbeta, float* dbias) {
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] = (float)sum_dzx[c];
      dbeta[c] = (float)sum_dz[c];
    }
  }
  const double invM = 1.0 / (double)count;       // pixels the sums were taken over (all ranks under SyncBN)
  const int c0t = (threadIdx.x % (C / 8)) * 8;
  float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[c0t + j]; sh[j] = shift[c0t + j]; mu[j] = mean[c0t + j]; is[j] = invstd[c0t + j];
    k1[j] = (float)(sum_dz[c0t + j] * invM);
    k2[j] = (float)(sum_dzx[c0t + j] * invM);
  }
  colreduce<T, float, 1, 1>(M, C,
                         [&](long m, int c0, float (&q0)[8], float (&q1)[8]) {
                           float g[8], yv[8], o[8];
                           load8(da + m * da_ld + c0, g);
                           load8(y + m * y_ld + c0, yv);
#pragma unroll
                           for (int j = 0; j < 8; ++j) {
                             o[j] = bn_bwd_dy(yv[j], g[j], sc[j], sh[j], mu[j], is[j], k1[j], k2[j]);
                             q0[j] = o[j];
                           }
                           store8(dy + m * dy_ld + c0, o);
                         },
                         dbias, (float*)nullptr);
}

// The same without the (never used) pre-BatchNorm bias gradient: a pure streaming kernel like bn_act_kernel --
// thread-fixed channel chunk, constants in registers, U rows in flight, no LDS, no reduction tail.
// REP: sum_dz / sum_dzx are the producer's `nrep` replica accumulators [nrep][C] (no sum_replicas launch in between):
// every workgroup adds them up itself (replica_sums_block), workgroup 0 writes the gamma / beta gradients
#ifndef CRIMAC_BNB_APPLY_WAVES
#define CRIMAC_BNB_APPLY_WAVES 2     // waves per SIMD the kernel is compiled for (4: at most 128 registers, two rows in flight)
#endif
// Tried (CRIMAC_BNB_APPLY_WAVES=4): at most 128 registers, so that a wave of this kernel fits BESIDE the two waves per SIMD
// of the plane-pair weight-gradient kernel (2 x 192 registers; it needs no LDS) and this HBM-bound pass could run on the same
// CUs as the MFMA-bound weight gradient of the previous block on the two-stream step: 26.25 vs 26.13 ms -- no gain, the
// streams do not put the two kernels on the chip at the same time often enough.
template <typename T, typename TD = T, bool REP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CRIMAC_BNB_APPLY_WAVES)))
void bn_bwd_apply_stream_kernel(
    const T* __restrict__ da, long da_ld, const T* __restrict__ y, long y_ld, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const double* __restrict__ sum_dz, const double* __restrict__ sum_dzx, long M, long count, int C,
    TD* __restrict__ dy, long dy_ld, float* dgamma, float* dbeta, int nrep) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
  float* k12 = reinterpret_cast<float*>(bn_smem + 512 * sizeof(double));      // [2][C]
  const double invM = 1.0 / (double)count;
  if constexpr (REP) {
    const bool writer = blockIdx.x == 0;
    replica_sums_block(sum_dz, sum_dzx, nrep, C, reinterpret_cast<double*>(bn_smem), [&](int c, double s1, double s2) {
      k12[c] = (float)(s1 * invM);
      k12[C + c] = (float)(s2 * invM);
      if (writer) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
    });
    __syncthreads();
  } else if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] = (float)sum_dzx[c];
      dbeta[c] = (float)sum_dz[c];
    }
  }
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  if (rl >= rpi) return;
  const int c0 = chunk * 8;
  float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
    if constexpr (REP) {
      k1[j] = k12[c0 + j];
      k2[j] = k12[C + c0 + j];
    } else {
      k1[j] = (float)(sum_dz[c0 + j] * invM);
      k2[j] = (float)(sum_dzx[c0 + j] * invM);
    }
  }
  const long stride = (long)gridDim.x * rpi;
#ifdef CRIMAC_BNB_APPLY_U
  constexpr int U = CRIMAC_BNB_APPLY_U;
#else
  constexpr int U = CRIMAC_BNB_APPLY_WAVES >= 4 ? 2 : 4;      // rows in flight per thread (two fit the 128-register budget)
#endif
  for (long m = (long)blockIdx.x * rpi + rl; m < M; m += U * stride) {
    float g[U][8], yv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (m + u * stride < M) {
        load8s(da + CRIMAC_ROW(m + u * stride, M) * da_ld + c0, g[u]);
        load8s(y + CRIMAC_ROW(m + u * stride, M) * y_ld + c0, yv[u]);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (m + u * stride >= M) continue;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = bn_bwd_dy(yv[u][j], g[u][j], sc[j], sh[j], mu[j], is[j], k1[j], k2[j]);
      store8(dy + CRIMAC_ROW(m + u * stride, M) * dy_ld + c0, o);
    }
  }
}

// ---- max-pool / skip backward + BatchNorm + ReLU backward in one pass (round 4) ----------------------------------------
// The block output of an encoder level feeds the max-pool and the skip connection: its gradient da = ds + unpool(dp) used
// to be WRITTEN by unpool_add (which takes the BatchNorm-backward sums on the way) and READ again by bn_bwd_apply.  Here
// unpool_add runs sums-only (da == nullptr) and this kernel rebuilds da from (dp, ds, y) -- the same arithmetic, the same
// rounding to the storage type of da -- while it forms dy: 11 instead of 12.5 bytes per element in bf16 (20 instead of 23
// in h3f), one tensor less in HBM.  T: storage of dp, ds, y (and of the da that is no longer stored); TA: of the forward
// activation (decides the rounding the pool argmax saw); TD: of dy.
template <typename T, typename TD, typename TA>
__global__ __launch_bounds__(256) void unpool_bn_bwd_apply_kernel(
    const T* __restrict__ dp, long dp_ld, const T* __restrict__ ds, long ds_ld, const T* __restrict__ y, long y_ld,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const double* __restrict__ sum_dz, const double* __restrict__ sum_dzx, int nrep,
    long count, TD* __restrict__ dy, long dy_ld, int B, int H, int W, int C, float* dgamma, float* dbeta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
  float* k12 = reinterpret_cast<float*>(bn_smem + 512 * sizeof(double));      // [2][C]
  const double invM = 1.0 / (double)count;
  {
    const bool writer = blockIdx.x == 0;
    replica_sums_block(sum_dz, sum_dzx, nrep, C, reinterpret_cast<double*>(bn_smem), [&](int c, double s1, double s2) {
      k12[c] = (float)(s1 * invM);
      k12[C + c] = (float)(s2 * invM);
      if (writer) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
    });
    __syncthreads();
  }
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  if (rl >= rpi) return;
  const int c0 = chunk * 8;
  float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
    k1[j] = k12[c0 + j];
    k2[j] = k12[C + c0 + j];
  }
  const int Hp = H / 2, Wp = W / 2;
  const long npool = (long)B * Hp * Wp;
  for (long r = (long)blockIdx.x * rpi + rl; r < npool; r += (long)gridDim.x * rpi) {
    const int xp = (int)(r % Wp);
    const long t = r / Wp;
    const int yp = (int)(t % Hp);
    const long b = t / Hp;
    float g[8], av[4][8], yv[4][8];
    load8s(dp + r * dp_ld + c0, g);
    long pix[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      pix[d] = (b * H + 2 * yp + (d >> 1)) * (long)W + 2 * xp + (d & 1);
      load8s(y + pix[d] * y_ld + c0, yv[d]);
#pragma unroll
      for (int j = 0; j < 8; ++j) av[d][j] = storage_round<TA>(fmaxf(yv[d][j] * sc[j] + sh[j], 0.f));   // (as unpool_add)
    }
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int best = 0;
      float bv = av[0][j];
#pragma unroll
      for (int d = 1; d < 4; ++d)
        if (av[d][j] > bv) { bv = av[d][j]; best = d; }   // first maximum wins (aten max_pool2d)
      arg[j] = best;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float o[8];
      if (ds) load8s(ds + pix[d] * ds_ld + c0, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dav = storage_round<T>((ds ? o[j] : 0.f) + (arg[j] == d ? g[j] : 0.f));     // da as it was stored
        o[j] = bn_bwd_dy(yv[d][j], dav, sc[j], sh[j], mu[j], is[j], k1[j], k2[j]);
      }
      store8(dy + pix[d] * dy_ld + c0, o);
    }
  }
}

// ---- 1x1 head --------------------------------------------------------------------------------------------
// T: storage type of x; TR: the type the activation formed on the fly (bn_scale given) is rounded to -- what bn_act
// would have stored (H3P: x = fp32 conv output, activation = fp16 plane pairs)
template <typename T, int NC, typename TR = T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, long x_ld, int Cin,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ logits, long npix, long HW,
                                                       int softmax, const float* __restrict__ bn_scale,
                                                       const float* __restrict__ bn_shift) {
  const int lp = Cin / 8;               // lanes per pixel (power of two <= 64)
  const int sub = threadIdx.x % lp;
  // bn_scale: x is the raw conv output of the last BatchNorm block, its activation relu(x*scale+shift)
  // (rounded to the storage type, as bn_act would have stored it) is formed here -- one pass over it saved
  float bsc[8], bsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bsc[j] = bn_scale ? bn_scale[sub * 8 + j] : 1.f;
    bsh[j] = bn_scale ? bn_shift[sub * 8 + j] : 0.f;
  }
  float wr[NC][8];
#pragma unroll
  for (int o = 0; o < NC; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = w[o * Cin + sub * 8 + j];
  const long ppb = 256 / lp;            // pixels per block-iteration
  // (one pixel per thread and iteration: with several unrolled copies a pixel's sums depended, in the last bit, on
  // which copy it met -- i.e. on the batch size; patches must not see each other)
  for (long p0 = (long)blockIdx.x * ppb; p0 < npix; p0 += (long)gridDim.x * ppb) {
    const long p = p0 + threadIdx.x / lp;
    float acc[NC];
#pragma unroll
    for (int o = 0; o < NC; ++o) acc[o] = 0.f;
    if (p < npix) {
      float v[8];
      load8(x + p * x_ld + sub * 8, v);
      if (bn_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = storage_round<TR>(fmaxf(v[j] * bsc[j] + bsh[j], 0.f));
      }
#pragma unroll
      for (int o = 0; o < NC; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o] = __builtin_fmaf(v[j], wr[o][j], acc[o]);
    }
    for (int off = lp >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int o = 0; o < NC; ++o) acc[o] += __shfl_xor(acc[o], off, 64);
    if (sub == 0 && p < npix) {
      float z[NC];
#pragma unroll
      for (int o = 0; o < NC; ++o) z[o] = acc[o] + bias[o];
      if (softmax) {
        float mx = z[0];
#pragma unroll
        for (int o = 1; o < NC; ++o) mx = fmaxf(mx, z[o]);
        float den = 0.f;
#pragma unroll
        for (int o = 0; o < NC; ++o) { z[o] = expf(z[o] - mx); den += z[o]; }
#pragma unroll
        for (int o = 0; o < NC; ++o) z[o] /= den;
      }
      long b, hw;
      pix_split(p, HW, b, hw);
#pragma unroll
      for (int o = 0; o < NC; ++o) logits[(b * NC + o) * HW + hw] = z[o];
    }
  }
}

// Cin == 64 (the configured net): a group of 8 lanes takes 8 consecutive pixels.  Lane s loads channel chunk s of each
// of them (every load instruction of the wave covers whole 128-byte pixel rows) and forms the partial dot products;
// a TRANSPOSING butterfly (xor 4, 2, 1: each step hands half of the pixels to the partner lane) leaves lane s with
// the complete sums of pixel s -- the same pairing, hence bit for bit the same sums, as the all-reduce butterfly of
// the generic kernel above, with 21 shuffles per 8 pixels instead of 72, the softmax evaluated once per pixel instead
// of in 8 lanes of which 7 are discarded, and the 64 pixels of a wave stored as 256 contiguous bytes per class plane.
template <typename T, int NC, typename TR = T>
__global__ __launch_bounds__(256) void head_fwd64_kernel(const T* __restrict__ x, long x_ld, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ logits,
                                                         long npix, long HW, int softmax,
                                                         const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift) {
  const int sub = threadIdx.x & 7;
  float bsc[8], bsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bsc[j] = bn_scale ? bn_scale[sub * 8 + j] : 1.f;
    bsh[j] = bn_scale ? bn_shift[sub * 8 + j] : 0.f;
  }
  float wr[NC][8];
#pragma unroll
  for (int o = 0; o < NC; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = w[o * 64 + sub * 8 + j];
  float bz[NC];
#pragma unroll
  for (int o = 0; o < NC; ++o) bz[o] = bias[o];
  const bool b4 = (sub & 4) != 0, b2 = (sub & 2) != 0, b1 = (sub & 1) != 0;
  const long step = (long)gridDim.x * 256;
  for (long g0 = (long)blockIdx.x * 256 + (threadIdx.x & ~7); g0 < npix; g0 += step) {
    float a[8][NC];
    float v[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (g0 + u < npix) load8s(x + (g0 + u) * x_ld + sub * 8, v[u]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (bn_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = storage_round<TR>(fmaxf(v[u][j] * bsc[j] + bsh[j], 0.f));
      }
#pragma unroll
      for (int o = 0; o < NC; ++o) {
        float s_ = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s_ = __builtin_fmaf(v[u][j], wr[o][j], s_);     // (explicit fma: left to contraction, the third class came out 1 ulp different between unrolled copies)
        a[u][o] = s_;
      }
    }
    float r4[4][NC], r2[2][NC], z[NC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int o = 0; o < NC; ++o) {
        const float mine = b4 ? a[u + 4][o] : a[u][o], theirs = b4 ? a[u][o] : a[u + 4][o];
        r4[u][o] = mine + __shfl_xor(theirs, 4, 64);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int o = 0; o < NC; ++o) {
        const float mine = b2 ? r4[u + 2][o] : r4[u][o], theirs = b2 ? r4[u][o] : r4[u + 2][o];
        r2[u][o] = mine + __shfl_xor(theirs, 2, 64);
      }
#pragma unroll
    for (int o = 0; o < NC; ++o) {
      const float mine = b1 ? r2[1][o] : r2[0][o], theirs = b1 ? r2[0][o] : r2[1][o];
      z[o] = mine + __shfl_xor(theirs, 1, 64) + bz[o];
    }
    const long p = g0 + sub;
    if (p < npix) {
      if (softmax) {
        float mx = z[0];
#pragma unroll
        for (int o = 1; o < NC; ++o) mx = fmaxf(mx, z[o]);
        float den = 0.f;
#pragma unroll
        for (int o = 0; o < NC; ++o) { z[o] = expf(z[o] - mx); den += z[o]; }
#pragma unroll
        for (int o = 0; o < NC; ++o) z[o] /= den;
      }
      long b, hw;
      pix_split(p, HW, b, hw);
#pragma unroll
      for (int o = 0; o < NC; ++o) logits[(b * NC + o) * HW + hw] = z[o];
    }
  }
}

// T: storage type of dx and of the y read for the fused sums; TX: of the head input x / of the activation rebuilt
// from y (H3P: fp16 plane pairs)
template <typename T, int NC, bool BNB, typename TX = T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl,
                                                       const TX* __restrict__ x, long x_ld, int Cin,
                                                       const float* __restrict__ w, T* __restrict__ dx,
                                                       long dx_ld, float* dw, float* db, long npix,
                                                       long HW, BnbArgs bnb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);   // [NC][Cin] + [NC]  (+ [2][Cin] for BNB)
  float* lds_bnb = lds + NC * Cin + NC;
  for (int i = threadIdx.x; i < NC * Cin + NC + (BNB ? 2 * Cin : 0); i += 256) lds[i] = 0.f;
  __syncthreads();
  const int lp = Cin / 8;
  const int sub = threadIdx.x % lp;
  BnbConst k;
  float d1[8], d2[8];
  if constexpr (BNB) {
    bnb_load(bnb, sub * 8, k);
#pragma unroll
    for (int j = 0; j < 8; ++j) { d1[j] = 0.f; d2[j] = 0.f; }
  }
  float wr[NC][8], gw[NC][8], gb[NC];
#pragma unroll
  for (int o = 0; o < NC; ++o) {
    gb[o] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { wr[o][j] = w[o * Cin + sub * 8 + j]; gw[o][j] = 0.f; }
  }
  const long ppb = 256 / lp;
  // Two pixels per iteration, every load (x, y of the BatchNorm sums, dlogits) issued before the first use:
  // with one pixel and the y load behind the dx store the loop was two dependent memory round trips per
  // pixel.  (image, pixel-in-image) advance incrementally -- no 64-bit divide per pixel.
  constexpr int UNR = 2;      // (three: 228 registers, no faster; four: one wave per SIMD)
  const long step = (long)gridDim.x * ppb;
  const long step_b = (UNR * step) / HW, step_hw = (UNR * step) % HW;
  long p = (long)blockIdx.x * ppb + threadIdx.x / lp;
  long pb[UNR], phw[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) { pb[u] = (p + u * step) / HW; phw[u] = (p + u * step) % HW; }
  for (; p < npix; p += UNR * step) {
    float g[UNR][NC], v[UNR][8], yv[UNR][8];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long pu = p + u * step;
      ok[u] = pu < npix;
      if (phw[u] >= HW) { phw[u] -= HW; pb[u] += 1; }
      if (ok[u]) {
#pragma unroll
        for (int o = 0; o < NC; ++o) g[u][o] = dl[(pb[u] * NC + o) * HW + phw[u]];
        if (x) load8s(x + pu * x_ld + sub * 8, v[u]);
        if constexpr (BNB) load8s(reinterpret_cast<const T*>(bnb.y) + pu * bnb.y_ld + sub * 8, yv[u]);
      }
      pb[u] += step_b;
      phw[u] += step_hw;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!ok[u]) continue;
      const long pu = p + u * step;
      float o8[8];
      if constexpr (BNB) {
        if (!x) {        // the head input is the activation of the block whose y is at hand: rebuild it
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = storage_round<TX>(fmaxf(yv[u][j] * k.sc[j] + k.sh[j], 0.f));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < NC; ++o) { s += g[u][o] * wr[o][j]; gw[o][j] += g[u][o] * v[u][j]; }
        o8[j] = s;
      }
      if (sub == 0)
#pragma unroll
        for (int o = 0; o < NC; ++o) gb[o] += g[u][o];
      if constexpr (BNB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = storage_round<T>(o8[j]);      // the sums see dx as it is stored
      }
      store8(dx + pu * dx_ld + sub * 8, o8);
      if constexpr (BNB) bnb_accum_v(k, yv[u], o8, d1, d2);
    }
  }
#pragma unroll
  for (int o = 0; o < NC; ++o) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&lds[o * Cin + sub * 8 + j], gw[o][j]);
    if (sub == 0) atomicAdd(&lds[NC * Cin + o], gb[o]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NC * Cin; i += 256) atomicAdd(&dw[i], lds[i]);
  if (threadIdx.x < NC) atomicAdd(&db[threadIdx.x], lds[NC * Cin + threadIdx.x]);
  if constexpr (BNB) bnb_flush(bnb, lds_bnb, Cin, sub * 8, d1, d2, 256);
}

// ---- weighted cross entropy ---------------------------------------------------------------------------------
__device__ __forceinline__ long load_label(const void* labels, int nbytes, long i) {
  if (nbytes == 8) return ((const long long*)labels)[i];
  if (nbytes == 4) return ((const int*)labels)[i];
  return ((const short*)labels)[i];
}

template <int NC>
__global__ __launch_bounds__(256) void wce_fwd_kernel(const float* __restrict__ logits,
                                                      const void* __restrict__ labels, int lbytes,
                                                      const float* __restrict__ cw, int ignore,
                                                      long npix, long HW, double* sums) {
  __shared__ double red[2][4];
  double s0 = 0.0, s1 = 0.0;
  auto pixel = [&](long y, const float (&z)[NC]) {
    if (y == ignore) return;
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < NC; ++o) mx = fmaxf(mx, z[o]);
    float den = 0.f, zy = 0.f, wy = 0.f;
#pragma unroll
    for (int o = 0; o < NC; ++o) {
      den += expf(z[o] - mx);
      if (o == y) { zy = z[o]; wy = cw[o]; }
    }
    const float nll = (mx + logf(den)) - zy;
    s0 += (double)(wy * nll);
    s1 += (double)wy;
  };
  if ((HW & 3) == 0 && (reinterpret_cast<unsigned long>(logits) & 15) == 0) {
    // four pixels of one image per thread and iteration: one 16-byte load per class plane (the one-pixel form below was a
    // chain of 4-byte loads behind the label test: 36 us for 29 MB at B = 32)
    const long nq = npix >> 2;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
      const long p = q << 2;
      long b, hw;
      pix_split(p, HW, b, hw);
      f32x4 zv[NC];
#pragma unroll
      for (int o = 0; o < NC; ++o) zv[o] = *reinterpret_cast<const f32x4*>(logits + (b * NC + o) * HW + hw);
      long yl[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) yl[k] = load_label(labels, lbytes, p + k);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float z[NC];
#pragma unroll
        for (int o = 0; o < NC; ++o) z[o] = zv[o][k];
        pixel(yl[k], z);
      }
    }
  } else {
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
      const long y = load_label(labels, lbytes, p);
      if (y == ignore) continue;
      long b, hw;
      pix_split(p, HW, b, hw);
      float z[NC];
#pragma unroll
      for (int o = 0; o < NC; ++o) z[o] = logits[(b * NC + o) * HW + hw];
      pixel(y, z);
    }
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = s0; red[1][wv] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

template <int NC>
__global__ __launch_bounds__(256) void wce_bwd_kernel(const float* __restrict__ logits,
                                                      const void* __restrict__ labels, int lbytes,
                                                      const float* __restrict__ cw, int ignore,
                                                      long npix, long HW, const double* sums,
                                                      float upstream, float* __restrict__ dl) {
  const float inv = upstream / (float)sums[1];
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix;
       p += (long)gridDim.x * blockDim.x) {
    const long y = load_label(labels, lbytes, p);
    long b, hw;
    pix_split(p, HW, b, hw);
    float z[NC];
    if (y == ignore) {
#pragma unroll
      for (int o = 0; o < NC; ++o) dl[(b * NC + o) * HW + hw] = 0.f;
      continue;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < NC; ++o) { z[o] = logits[(b * NC + o) * HW + hw]; mx = fmaxf(mx, z[o]); }
    float den = 0.f, wy = 0.f;
#pragma unroll
    for (int o = 0; o < NC; ++o) {
      z[o] = expf(z[o] - mx);
      den += z[o];
      if (o == y) wy = cw[o];
    }
    const float k = wy * inv;
#pragma unroll
    for (int o = 0; o < NC; ++o)
      dl[(b * NC + o) * HW + hw] = k * (z[o] / den - (o == y ? 1.f : 0.f));
  }
}

// softmax over the class axis of NCHW logits (F.softmax(dim=1), pipeline.py:218, :269): one pixel per thread
template <int NC>
__global__ __launch_bounds__(256) void softmax_nchw_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                           long npix, long HW) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    long b, hw;
    pix_split(p, HW, b, hw);
    float z[NC];
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < NC; ++o) { z[o] = logits[(b * NC + o) * HW + hw]; mx = fmaxf(mx, z[o]); }
    float den = 0.f;
#pragma unroll
    for (int o = 0; o < NC; ++o) { z[o] = expf(z[o] - mx); den += z[o]; }      // (same arithmetic as the head's fused softmax)
#pragma unroll
    for (int o = 0; o < NC; ++o) out[(b * NC + o) * HW + hw] = z[o] / den;
  }
}

// ---- SGD with momentum over a flat parameter buffer -------------------------------------------------------------
// Loss-scaled training (fp16 storage): state[0] = 1 if any gradient is not finite (this step), state[1] counts
// the steps skipped because of it.  The check is a pass of its own over the flat gradient (31 M floats, ~25 us).
__global__ __launch_bounds__(256) void grad_overflow_kernel(const float* __restrict__ g, long n4, long n, int* state) {
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    // |x| <= FLT_MAX is false for inf and NaN
    bad |= !(fabsf(v[0]) <= 3.402823466e38f) | !(fabsf(v[1]) <= 3.402823466e38f) | !(fabsf(v[2]) <= 3.402823466e38f) |
           !(fabsf(v[3]) <= 3.402823466e38f);
  }
  const long t = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t < n) bad |= !(fabsf(g[t]) <= 3.402823466e38f);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&state[0], 1);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ v, long n4, long n, float lr,
                                                  float mom, float gscale, int zero_grad, int* guard) {
  if (guard && guard[0]) {       // non-finite gradients: skip the update (all threads see the same flag)
    if (blockIdx.x == 0 && threadIdx.x == 0) guard[1] += 1;
    return;
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
#if CRIMAC_STREAM_NT      // (gradient and velocity are touched once per step: streamed past the caches)
    f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
#else
    f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vv[j] = mom * vv[j] + gv[j] * gscale;
      pv[j] -= lr * vv[j];
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
#if CRIMAC_STREAM_NT
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
#else
    reinterpret_cast<f32x4*>(v)[i] = vv;
#endif
    if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // tail (n not a multiple of 4)
  const long tail0 = n4 * 4;
  const long t = tail0 + blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t < n) {
    const float vv = mom * v[t] + g[t] * gscale;
    v[t] = vv;
    p[t] -= lr * vv;
    if (zero_grad) g[t] = 0.f;
  }
}

}  // namespace

// =================================================================================================
//                                         C ABI
// =================================================================================================
#define ST ((hipStream_t)stream)
#define PREC_OK(name) \
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, name ": bad precision %d", prec)
// ... and the entry points that also take CRIMAC_PREC_H3F_BWD (da, y fp32 -> dy fp16)
#define PREC_OK_BWD16(name) \
  CRIMAC_REQUIRE((prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX) || prec == CRIMAC_PREC_H3F_BWD, name ": bad precision %d", prec)

extern "C" int crimac_nchw_to_nhwc(int prec, const float* in, void* out, int B, int C, int H, int W,
                                   long ld, void* stream) {
  PREC_OK("nchw_to_nhwc");
  CRIMAC_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && ld >= C && ld % 8 == 0,
                 "nchw_to_nhwc: bad arguments (C=%d ld=%ld)", C, ld);
  const long npix = (long)B * H * W;
  const int grid = grid_for(npix, 256);
  CRIMAC_FOR_STORAGE2(prec, TF, T, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid), dim3(256), 0, ST, in, (T*)out, C,
                                                      (long)H * W, npix, ld));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_pack_conv3x3(const float* w, int Co, int Ci, int Ci_pad, const float* scale,
                                   int planes, void* fwd_hi, void* fwd_lo, void* dg_hi, void* dg_lo,
                                   void* stream) {
  CRIMAC_REQUIRE(w && fwd_hi && Co > 0 && Ci > 0 && Ci_pad >= Ci, "pack_conv3x3: bad arguments");
  CRIMAC_REQUIRE(planes_arg_ok(planes) && ((planes & 15) == 1 || (planes & CRIMAC_PLANES_INTERLEAVED) || fwd_lo) &&
                     ((planes & 15) == 1 || (planes & CRIMAC_PLANES_INTERLEAVED) || !dg_hi || dg_lo),
                 "pack_conv3x3: planes=%d needs the lo plane buffers", planes);
  CRIMAC_REQUIRE(!dg_hi || Ci_pad == Ci, "pack_conv3x3: dgrad planes need Ci_pad == Ci");
  CRIMAC_REQUIRE(!(planes & CRIMAC_PLANES_FWD_FRAG) ||
                     (Co % 32 == 0 && (((planes & CRIMAC_PLANES_INTERLEAVED) && Ci_pad % 32 == 0) ||
                                       ((planes & 15) == 1 && Ci_pad % 64 == 0))),
                 "pack_conv3x3: a fragment-major forward plane needs Co %% 32 == 0 and Ci_pad %% 64 == 0 (single 16-bit plane) or "
                 "Ci_pad %% 32 == 0 (interleaved pairs) (Co=%d Ci_pad=%d planes=%d)", Co, Ci_pad, planes);
  const int grid = grid_for(9L * Co * Ci_pad, 256);
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid), dim3(256), 0, ST, w, Co, Ci, Ci_pad, scale, planes,
                     (unsigned short*)fwd_hi, (unsigned short*)fwd_lo, (unsigned short*)dg_hi,
                     (unsigned short*)dg_lo);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_pack_upconv2x2(const float* w, int Ci, int Co, int planes, void* fwd_hi,
                                     void* fwd_lo, void* dg_hi, void* dg_lo, void* stream) {
  CRIMAC_REQUIRE(w && fwd_hi && Co > 0 && Ci > 0, "pack_upconv2x2: bad arguments");
  CRIMAC_REQUIRE(planes_arg_ok(planes) && ((planes & 15) == 1 || (planes & CRIMAC_PLANES_INTERLEAVED) || fwd_lo) &&
                     ((planes & 15) == 1 || (planes & CRIMAC_PLANES_INTERLEAVED) || !dg_hi || dg_lo),
                 "pack_upconv2x2: planes=%d needs the lo plane buffers", planes);
  const int grid = grid_for(4L * Co * Ci, 256);
  hipLaunchKernelGGL(pack_upconv_kernel, dim3(grid), dim3(256), 0, ST, w, Ci, Co, planes,
                     (unsigned short*)fwd_hi, (unsigned short*)fwd_lo, (unsigned short*)dg_hi,
                     (unsigned short*)dg_lo);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_unpack_wgrad_conv3x3(const float* dw, int Co, int Ci, int Ci_pad, float* grad,
                                           void* stream) {
  CRIMAC_REQUIRE(dw && grad && Co > 0 && Ci > 0 && Ci_pad >= Ci, "unpack_wgrad_conv3x3: bad arguments");
  hipLaunchKernelGGL(unpack_wgrad_conv_kernel, dim3(grid_for(9L * Co * Ci, 256)), dim3(256), 0, ST, dw,
                     Co, Ci, Ci_pad, grad);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_unpack_wgrad_upconv2x2(const float* dw, int Ci, int Co, float* grad,
                                             void* stream) {
  CRIMAC_REQUIRE(dw && grad && Co > 0 && Ci > 0, "unpack_wgrad_upconv2x2: bad arguments");
  hipLaunchKernelGGL(unpack_wgrad_upconv_kernel, dim3(grid_for(4L * Co * Ci, 256)), dim3(256), 0, ST,
                     dw, Ci, Co, grad);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

static int colreduce_grid(long M, int C) {
  const int cpr = C / 8;
  const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;
  return grid_for(M, rpi * 16);
}

extern "C" int crimac_colstats(int prec, const void* y, long ld, long M, int C, double* sum,
                               double* sumsq, void* stream) {
  PREC_OK("colstats");
  CRIMAC_REQUIRE(y && sum && M > 0 && C > 0 && C % 8 == 0 && C <= 2048 && ld >= C && ld % 8 == 0,
                 "colstats: bad arguments (C=%d ld=%ld)", C, ld);
  const int grid = colreduce_grid(M, C);
  const size_t lds = 2 * C * sizeof(float);
  if (sumsq)
    CRIMAC_FOR_STORAGE(prec, T, hipLaunchKernelGGL((colstats_kernel<T, double, 2>), dim3(grid), dim3(256), lds, ST,
                                                   (const T*)y, ld, M, C, sum, sumsq));
  else
    CRIMAC_FOR_STORAGE(prec, T, hipLaunchKernelGGL((colstats_kernel<T, double, 1>), dim3(grid), dim3(256), lds, ST,
                                                   (const T*)y, ld, M, C, sum, sumsq));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_colsum_f32(int prec, const void* y, long ld, long M, int C, float* sum,
                                 void* stream) {
  PREC_OK("colsum_f32");
  CRIMAC_REQUIRE(y && sum && M > 0 && C > 0 && C % 8 == 0 && C <= 2048 && ld >= C && ld % 8 == 0,
                 "colsum_f32: bad arguments (C=%d ld=%ld)", C, ld);
  const int grid = colreduce_grid(M, C);
  const size_t lds = 2 * C * sizeof(float);
  CRIMAC_FOR_STORAGE(prec, T, hipLaunchKernelGGL((colstats_kernel<T, float, 1>), dim3(grid), dim3(256), lds, ST,
                                                 (const T*)y, ld, M, C, sum, (float*)nullptr));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_sum_replicas(const double* src, int replicas, long stride, int n, double* dst_f64,
                                   float* dst_f32, const double* src_b, double* dst_b_f64, void* stream) {
  CRIMAC_REQUIRE(src && replicas > 0 && n > 0 && stride >= n && (dst_f64 || dst_f32) && (!src_b == !dst_b_f64),
                 "sum_replicas: bad arguments");
  hipLaunchKernelGGL(sum_replicas_kernel, dim3(cdiv(n, 16), src_b ? 2 : 1), dim3(256), 0, ST, src, replicas,
                     stride, n, dst_f64, dst_f32, src_b, dst_b_f64);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_bn_finalize(const double* sum, const double* sumsq, int replicas, long M, int C,
                                  const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var,
                                  long long* num_batches_tracked, float* mean, float* invstd,
                                  float* scale, float* shift, void* stream) {
  CRIMAC_REQUIRE(sum && sumsq && gamma && beta && mean && invstd && scale && shift && M > 0 && C > 0 &&
                     replicas >= 1,
                 "bn_finalize: bad arguments");
  CRIMAC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, ST, sum, sumsq, replicas, M, C, gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean, invstd,
                     scale, shift);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

static int colreduce_grid(long M, int C);

template <typename T, typename TO = T, bool FIN = false>
static int bn_act_pool_launch(const void* y, long y_ld, const float* scale, const float* shift,
                              int relu, void* out, long out_ld, void* pool_out, long pool_ld, int B,
                              int H, int W, int C, hipStream_t st, BnFin fin = BnFin{}) {
  const size_t lds = FIN ? 512 * sizeof(double) + 4 * (size_t)C * sizeof(float) : 0;
  if (pool_out) {
    const long npool = (long)B * (H / 2) * (W / 2);
    const int cpr = C / 8, rpi = 256 / cpr > 0 ? 256 / cpr : 1;      // (pooled pixels per workgroup and iteration, as in the kernel)
    hipLaunchKernelGGL((bn_act_pool_kernel<T, TO, FIN>), dim3(whole_rounds(bn_act_pool_kernel<T, TO, FIN>, lds, grid_for(npool, rpi * 2))), dim3(256), lds, st,
                       (const T*)y, y_ld, scale, shift, relu, (TO*)out, out_ld, (TO*)pool_out, pool_ld, B,
                       H, W, C, fin);
  } else {
    const long M = (long)B * H * W;
    hipLaunchKernelGGL((bn_act_kernel<T, TO, FIN>), dim3(whole_rounds(bn_act_kernel<T, TO, FIN>, lds, colreduce_grid(M, C))), dim3(256), lds, st,
                       (const T*)y, y_ld, scale, shift, relu, (TO*)out, out_ld, M, C, fin);
  }
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_bn_act_pool(int prec, const void* y, long y_ld, const float* scale,
                                  const float* shift, int relu, void* out, long out_ld,
                                  void* pool_out, long pool_ld, int B, int H, int W, int C,
                                  void* stream) {
  PREC_OK("bn_act_pool");
  CRIMAC_REQUIRE(y && (out || pool_out) && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "bn_act_pool: bad arguments");
  CRIMAC_REQUIRE((scale == nullptr) == (shift == nullptr), "bn_act_pool: scale/shift must come together");
  CRIMAC_REQUIRE(y_ld >= C && y_ld % 8 == 0 && (!out || (out_ld >= C && out_ld % 8 == 0)) &&
                     (!pool_out || (pool_ld >= C && pool_ld % 8 == 0)),
                 "bn_act_pool: bad pixel strides");
  CRIMAC_REQUIRE(!pool_out || (H % 2 == 0 && W % 2 == 0), "bn_act_pool: pooling needs even H, W");
  if (prec == CRIMAC_PREC_H3P) {
    // training: y = fp32 conv output (scale / shift given), activation and pooled tensor = plane pairs; inference
    // max-pool (no scale): plane pairs in and out
    if (scale) return bn_act_pool_launch<float, hp_t>(y, y_ld, scale, shift, relu, out, out_ld, pool_out, pool_ld, B, H, W, C, ST);
    return bn_act_pool_launch<hp_t, hp_t>(y, y_ld, scale, shift, relu, out, out_ld, pool_out, pool_ld, B, H, W, C, ST);
  }
  CRIMAC_FOR_STORAGE(prec, T, return bn_act_pool_launch<T>(y, y_ld, scale, shift, relu, out, out_ld, pool_out, pool_ld,
                                                           B, H, W, C, ST));
}

// Training-mode BatchNorm + ReLU (+ 2x2 max-pool) straight from the convolution epilogue's replica accumulators:
// crimac_bn_finalize and crimac_bn_act_pool in one launch (bn_fin_block above).
extern "C" int crimac_bn_train_act_pool(int prec, const void* y, long y_ld, const double* stat_sum,
                                        const double* stat_sumsq, int replicas, long count, const float* gamma,
                                        const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, long long* num_batches_tracked, float* bn_vec,
                                        long bn_stride, int relu, void* out, long out_ld, void* pool_out,
                                        long pool_ld, int B, int H, int W, int C, void* stream) {
  PREC_OK("bn_train_act_pool");
  CRIMAC_REQUIRE(y && (out || pool_out) && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "bn_train_act_pool: bad arguments");
  CRIMAC_REQUIRE(stat_sum && stat_sumsq && replicas >= 1 && count > 0 && gamma && beta && bn_vec && bn_stride >= C,
                 "bn_train_act_pool: bad statistics arguments");
  CRIMAC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_act_pool: running stats");
  CRIMAC_REQUIRE(y_ld >= C && y_ld % 8 == 0 && (!out || (out_ld >= C && out_ld % 8 == 0)) &&
                     (!pool_out || (pool_ld >= C && pool_ld % 8 == 0)),
                 "bn_train_act_pool: bad pixel strides");
  CRIMAC_REQUIRE(!pool_out || (H % 2 == 0 && W % 2 == 0), "bn_train_act_pool: pooling needs even H, W");
  const BnFin fin{stat_sum, stat_sumsq, replicas, count, gamma, beta, eps, momentum, running_mean, running_var,
                  num_batches_tracked, bn_vec, bn_stride};
  if (prec == CRIMAC_PREC_H3P)
    return bn_act_pool_launch<float, hp_t, true>(y, y_ld, nullptr, nullptr, relu, out, out_ld, pool_out, pool_ld, B, H, W,
                                                 C, ST, fin);
  CRIMAC_FOR_STORAGE(prec, T, return (bn_act_pool_launch<T, T, true>(y, y_ld, nullptr, nullptr, relu, out, out_ld, pool_out,
                                                                      pool_ld, B, H, W, C, ST, fin)));
}

static bool bnb_args_ok(const void* y, long y_ld, const float* vec, long stride, double* s0, double* s1, int replicas,
                        int C) {
  if (!s0) return true;                      // no fused reduction requested
  return y && vec && s1 && y_ld >= C && y_ld % 8 == 0 && stride >= C && replicas >= 1;
}

extern "C" int crimac_unpool_add(int prec, const void* dp, long dp_ld, const void* a, long a_ld,
                                 const void* ds, long ds_ld, void* da, long da_ld, int B, int H, int W,
                                 int C, const void* bnb_y, long bnb_y_ld, const float* bnb_vec, long bnb_stride,
                                 double* stat_sum, double* stat_sumsq, int stat_replicas, void* stream) {
  PREC_OK("unpool_add");
  CRIMAC_REQUIRE(dp && a && (da || stat_sum) && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 &&
                     C % 8 == 0 && C <= 2048,
                 "unpool_add: bad arguments (da may be NULL only with the fused BatchNorm-backward sums)");
  CRIMAC_REQUIRE(dp_ld >= C && a_ld >= C && (!da || da_ld >= C) && (!ds || ds_ld >= C) && dp_ld % 8 == 0 &&
                     a_ld % 8 == 0 && (!da || da_ld % 8 == 0) && (!ds || ds_ld % 8 == 0),
                 "unpool_add: bad pixel strides");
  CRIMAC_REQUIRE(bnb_args_ok(bnb_y, bnb_y_ld, bnb_vec, bnb_stride, stat_sum, stat_sumsq, stat_replicas, C),
                 "unpool_add: bad arguments of the fused BatchNorm-backward sums");
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  // with the fused sums every workgroup ends with 2 C LDS-atomic columns and 2 C fp64 atomics: four pooled pixels per
  // thread instead of one (512 ch @ 32^2: 61 -> see tools/bench_unpool.py)
  const int grid = grid_for(total, stat_sum ? 256 * 4 : 256);
  BnbArgs bnb{bnb_y, bnb_y_ld, bnb_vec, bnb_stride, stat_sum, stat_sumsq, stat_replicas};
#define UA(T, TA, BNB, LDS)                                                                               \
  hipLaunchKernelGGL((unpool_add_kernel<T, BNB, TA>), dim3(whole_rounds(unpool_add_kernel<T, BNB, TA>, LDS, grid)), dim3(256), LDS, ST, (const T*)dp, dp_ld, \
                     (const TA*)a, a_ld, (const T*)ds, ds_ld, (T*)da, da_ld, B, H, W, C, bnb)
  const size_t lds = 2 * (size_t)C * sizeof(float);
  if (stat_sum) CRIMAC_FOR_STORAGE2(prec, T, TA, UA(T, TA, true, lds));
  else CRIMAC_FOR_STORAGE2(prec, T, TA, UA(T, TA, false, 0));
#undef UA
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_bn_bwd_reduce(int prec, const void* da, long da_ld, const void* y, long y_ld,
                                    const float* scale, const float* shift, const float* mean,
                                    const float* invstd, long M, int C, double* sum_dz,
                                    double* sum_dz_xhat, void* stream) {
  PREC_OK("bn_bwd_reduce");
  CRIMAC_REQUIRE(da && y && scale && shift && mean && invstd && sum_dz && sum_dz_xhat && M > 0 &&
                     C > 0 && C % 8 == 0 && C <= 2048 && da_ld >= C && y_ld >= C && da_ld % 8 == 0 &&
                     y_ld % 8 == 0,
                 "bn_bwd_reduce: bad arguments");
  const int grid = colreduce_grid(M, C);
  const size_t lds = 2 * C * sizeof(float);
  CRIMAC_FOR_STORAGE(prec, T, hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(grid), dim3(256), lds, ST, (const T*)da,
                                                 da_ld, (const T*)y, y_ld, scale, shift, mean, invstd, M, C, sum_dz,
                                                 sum_dz_xhat));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_bn_bwd_apply(int prec, const void* da, long da_ld, const void* y, long y_ld,
                                   const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const double* sum_dz,
                                   const double* sum_dz_xhat, long M, long count, int C, void* dy, long dy_ld,
                                   float* dgamma, float* dbeta, float* dbias, void* stream) {
  PREC_OK_BWD16("bn_bwd_apply");
  CRIMAC_REQUIRE(da && y && scale && shift && mean && invstd && sum_dz && sum_dz_xhat && dy && dgamma &&
                     dbeta && M > 0 && C > 0 && C % 8 == 0 && C <= 2048 && (count == 0 || count >= M),
                 "bn_bwd_apply: bad arguments");
  if (count == 0) count = M;
  CRIMAC_REQUIRE(da_ld >= C && y_ld >= C && dy_ld >= C && da_ld % 8 == 0 && y_ld % 8 == 0 &&
                     dy_ld % 8 == 0,
                 "bn_bwd_apply: bad pixel strides");
  const int grid = colreduce_grid(M, C);
  const size_t lds = 2 * C * sizeof(float);
  static const int stream_form = getenv("CRIMAC_BNB_STREAM") ? atoi(getenv("CRIMAC_BNB_STREAM")) : 1;
  if (!dbias && stream_form && C <= 2048)
    CRIMAC_FOR_STORAGE2(prec, T, TD, hipLaunchKernelGGL((bn_bwd_apply_stream_kernel<T, TD>), dim3(whole_rounds(bn_bwd_apply_stream_kernel<T, TD>, 0, grid)), dim3(256), 0, ST,
                                                   (const T*)da, da_ld, (const T*)y, y_ld, scale, shift, mean, invstd,
                                                   sum_dz, sum_dz_xhat, M, count, C, (TD*)dy, dy_ld, dgamma, dbeta, 1));
  else
    CRIMAC_FOR_STORAGE2(prec, T, TD, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TD>), dim3(grid), dim3(256), lds, ST, (const T*)da,
                                                   da_ld, (const T*)y, y_ld, scale, shift, mean, invstd, sum_dz,
                                                   sum_dz_xhat, M, count, C, (TD*)dy, dy_ld, dgamma, dbeta, dbias));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// BatchNorm + ReLU backward straight from the producers' replica accumulators of (sum dz, sum dz*xhat):
// crimac_sum_replicas and crimac_bn_bwd_apply in one launch.  bn_vec: rows mean | invstd | scale | shift.
extern "C" int crimac_bn_bwd_apply_replicas(int prec, const void* da, long da_ld, const void* y, long y_ld,
                                            const float* bn_vec, long bn_stride, const double* sum_dz,
                                            const double* sum_dz_xhat, int replicas, long M, long count, int C,
                                            void* dy, long dy_ld, float* dgamma, float* dbeta, void* stream) {
  PREC_OK_BWD16("bn_bwd_apply_replicas");
  CRIMAC_REQUIRE(da && y && bn_vec && bn_stride >= C && sum_dz && sum_dz_xhat && replicas >= 1 && dy && dgamma &&
                     dbeta && M > 0 && C > 0 && C % 8 == 0 && C <= 2048 && (count == 0 || count >= M),
                 "bn_bwd_apply_replicas: bad arguments");
  if (count == 0) count = M;
  CRIMAC_REQUIRE(da_ld >= C && y_ld >= C && dy_ld >= C && da_ld % 8 == 0 && y_ld % 8 == 0 && dy_ld % 8 == 0,
                 "bn_bwd_apply_replicas: bad pixel strides");
  const int grid = colreduce_grid(M, C);
  const size_t lds = 512 * sizeof(double) + 2 * (size_t)C * sizeof(float);
  CRIMAC_FOR_STORAGE2(prec, T, TD, hipLaunchKernelGGL((bn_bwd_apply_stream_kernel<T, TD, true>), dim3(whole_rounds(bn_bwd_apply_stream_kernel<T, TD, true>, lds, grid)), dim3(256), lds,
                                                 ST, (const T*)da, da_ld, (const T*)y, y_ld, bn_vec + 2 * bn_stride,
                                                 bn_vec + 3 * bn_stride, bn_vec, bn_vec + bn_stride, sum_dz, sum_dz_xhat,
                                                 M, count, C, (TD*)dy, dy_ld, dgamma, dbeta, replicas));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// crimac_unpool_add (sums only: da == NULL) + crimac_bn_bwd_apply_replicas without the round trip of da through HBM.
// prec as for crimac_bn_bwd_apply_replicas (CRIMAC_PREC_H3F_BWD: dp, ds, y fp32 -> dy fp16; the pool argmax is taken on the
// plane-pair rounding of the activation, as the forward pass stored it).
extern "C" int crimac_unpool_bn_bwd_apply_replicas(int prec, const void* dp, long dp_ld, const void* ds, long ds_ld,
                                                   const void* y, long y_ld, const float* bn_vec, long bn_stride,
                                                   const double* sum_dz, const double* sum_dz_xhat, int replicas,
                                                   long count, void* dy, long dy_ld, int B, int H, int W, int C,
                                                   float* dgamma, float* dbeta, void* stream) {
  PREC_OK_BWD16("unpool_bn_bwd_apply_replicas");
  CRIMAC_REQUIRE(dp && y && bn_vec && bn_stride >= C && sum_dz && sum_dz_xhat && replicas >= 1 && dy && dgamma && dbeta &&
                     B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "unpool_bn_bwd_apply_replicas: bad arguments");
  const long M = (long)B * H * W;
  CRIMAC_REQUIRE(count == 0 || count >= M, "unpool_bn_bwd_apply_replicas: count smaller than the pixels of this call");
  if (count == 0) count = M;
  CRIMAC_REQUIRE(dp_ld >= C && y_ld >= C && dy_ld >= C && (!ds || ds_ld >= C) && dp_ld % 8 == 0 && y_ld % 8 == 0 &&
                     dy_ld % 8 == 0 && (!ds || ds_ld % 8 == 0),
                 "unpool_bn_bwd_apply_replicas: bad pixel strides");
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  // every workgroup adds up the replica accumulators first (replicas x C x 2 doubles): four pooled pixels = 16 rows per
  // thread, the share of bn_bwd_apply_stream_kernel's threads (one per thread: 80 us for 512 ch @ 32^2, the plain apply 36)
  const int grid = grid_for(total, 256 * 4);
  const size_t lds = 512 * sizeof(double) + 2 * (size_t)C * sizeof(float);
#define UBA(T, TD, TA)                                                                                                  \
  hipLaunchKernelGGL((unpool_bn_bwd_apply_kernel<T, TD, TA>), dim3(whole_rounds(unpool_bn_bwd_apply_kernel<T, TD, TA>, lds, grid)), dim3(256), lds, ST, (const T*)dp, dp_ld,      \
                     (const T*)ds, ds_ld, (const T*)y, y_ld, bn_vec + 2 * bn_stride, bn_vec + 3 * bn_stride, bn_vec,    \
                     bn_vec + bn_stride, sum_dz, sum_dz_xhat, replicas, count, (TD*)dy, dy_ld, B, H, W, C, dgamma, dbeta)
  if (prec == CRIMAC_PREC_H3F_BWD) UBA(float, half_t, hp_t);
  else CRIMAC_FOR_STORAGE2(prec, T, TP, UBA(T, TP, TP));
#undef UBA
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

template <typename T, typename TR = T>
static int head_fwd_launch(const void* x, long x_ld, int Cin, const float* w, const float* b,
                           float* logits, long npix, long HW, int ncls, int softmax, const float* bn_scale,
                           const float* bn_shift, hipStream_t st) {
  const int lp = Cin / 8;
  if (Cin == 64) {
    const int grid64 = grid_for(npix, 256 * 4);
#define HF64(NC)                                                                                              \
  hipLaunchKernelGGL((head_fwd64_kernel<T, NC, TR>), dim3(whole_rounds(head_fwd64_kernel<T, NC, TR>, 0, grid64)), dim3(256), 0, st, (const T*)x, x_ld, w, b, logits, \
                     npix, HW, softmax, bn_scale, bn_shift)
    if (ncls == 2) HF64(2); else if (ncls == 3) HF64(3); else HF64(4);
#undef HF64
    CRIMAC_LAUNCH_CHECK();
    return CRIMAC_OK;
  }
  const int grid = grid_for(npix, (256 / lp) * 8);
#define HF(NC)                                                                                      \
  hipLaunchKernelGGL((head_fwd_kernel<T, NC, TR>), dim3(whole_rounds(head_fwd_kernel<T, NC, TR>, 0, grid)), dim3(256), 0, st, (const T*)x, x_ld, Cin, \
                     w, b, logits, npix, HW, softmax, bn_scale, bn_shift)
  if (ncls == 2) HF(2); else if (ncls == 3) HF(3); else HF(4);
#undef HF
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

static bool head_cin_ok(int Cin) {
  const int lp = Cin / 8;
  return Cin > 0 && Cin % 8 == 0 && lp <= 64 && (lp & (lp - 1)) == 0;
}

extern "C" int crimac_head_fwd(int prec, const void* x, long x_ld, int Cin, const float* w,
                               const float* b, float* logits, int B, int H, int W, int ncls,
                               int softmax, const float* bn_scale, const float* bn_shift, void* stream) {
  PREC_OK("head_fwd");
  CRIMAC_REQUIRE(x && w && b && logits && B > 0 && H > 0 && W > 0, "head_fwd: bad arguments");
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "head_fwd: ncls=%d unsupported (2..4)", ncls);
  CRIMAC_REQUIRE(head_cin_ok(Cin) && x_ld >= Cin && x_ld % 8 == 0,
                 "head_fwd: Cin=%d must be 8*2^k <= 512", Cin);
  CRIMAC_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "head_fwd: bn_scale and bn_shift go together");
  const long HW = (long)H * W;
  if (prec == CRIMAC_PREC_H3P) {
    // x = the fp32 conv output of the last block (its activation is formed on the fly, rounded to plane pairs), or --
    // inference -- the plane-pair activation itself
    if (bn_scale) return head_fwd_launch<float, hp_t>(x, x_ld, Cin, w, b, logits, B * HW, HW, ncls, softmax, bn_scale, bn_shift, ST);
    return head_fwd_launch<hp_t, hp_t>(x, x_ld, Cin, w, b, logits, B * HW, HW, ncls, softmax, bn_scale, bn_shift, ST);
  }
  CRIMAC_FOR_STORAGE(prec, T, return head_fwd_launch<T>(x, x_ld, Cin, w, b, logits, B * HW, HW, ncls, softmax, bn_scale,
                                                        bn_shift, ST));
}

template <typename T, typename TX = T>
static int head_bwd_launch(const float* dl, const void* x, long x_ld, int Cin, const float* w, void* dx,
                           long dx_ld, float* dw, float* db, long npix, long HW, int ncls, const BnbArgs& bnb,
                           hipStream_t st) {
  const int lp = Cin / 8;
  int grid = grid_for(npix, (256 / lp) * 32);
  if (grid > 1024) grid = 1024;
  const size_t lds = (size_t)(ncls * Cin + ncls + (bnb.sum_dz ? 2 * Cin : 0)) * sizeof(float);
#define HB(NC, BNB)                                                                                          \
  hipLaunchKernelGGL((head_bwd_kernel<T, NC, BNB, TX>), dim3(whole_rounds(head_bwd_kernel<T, NC, BNB, TX>, lds, grid)), dim3(256), lds, st, dl, (const TX*)x, x_ld,   \
                     Cin, w, (T*)dx, dx_ld, dw, db, npix, HW, bnb)
  if (bnb.sum_dz) {
    if (ncls == 2) HB(2, true); else if (ncls == 3) HB(3, true); else HB(4, true);
  } else {
    if (ncls == 2) HB(2, false); else if (ncls == 3) HB(3, false); else HB(4, false);
  }
#undef HB
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_head_bwd(int prec, const float* dlogits, const void* x, long x_ld, int Cin,
                               const float* w, void* dx, long dx_ld, float* dw, float* db, int B, int H,
                               int W, int ncls, const void* bnb_y, long bnb_y_ld, const float* bnb_vec,
                               long bnb_stride, double* stat_sum, double* stat_sumsq, int stat_replicas,
                               void* stream) {
  PREC_OK("head_bwd");
  CRIMAC_REQUIRE(dlogits && w && dx && dw && db && B > 0 && H > 0 && W > 0, "head_bwd: bad arguments");
  CRIMAC_REQUIRE(x || (bnb_y && stat_sum), "head_bwd: x may only be NULL together with the fused BatchNorm sums");
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "head_bwd: ncls=%d unsupported (2..4)", ncls);
  CRIMAC_REQUIRE(head_cin_ok(Cin) && (!x || (x_ld >= Cin && x_ld % 8 == 0)) && dx_ld >= Cin && dx_ld % 8 == 0,
                 "head_bwd: Cin=%d must be 8*2^k <= 512", Cin);
  CRIMAC_REQUIRE(bnb_args_ok(bnb_y, bnb_y_ld, bnb_vec, bnb_stride, stat_sum, stat_sumsq, stat_replicas, Cin),
                 "head_bwd: bad arguments of the fused BatchNorm-backward sums");
  const long HW = (long)H * W;
  const BnbArgs bnb{bnb_y, bnb_y_ld, bnb_vec, bnb_stride, stat_sum, stat_sumsq, stat_replicas};
  CRIMAC_FOR_STORAGE2(prec, T, TX, return (head_bwd_launch<T, TX>(dlogits, x, x_ld, Cin, w, dx, dx_ld, dw, db, B * HW, HW, ncls,
                                                                    bnb, ST)));
}

extern "C" int crimac_wce_fwd(const float* logits, const void* labels, int label_bytes,
                              const float* class_w, int ncls, int ignore_index, int B, int H, int W,
                              double* sums, void* stream) {
  CRIMAC_REQUIRE(logits && labels && class_w && sums && B > 0 && H > 0 && W > 0, "wce_fwd: bad arguments");
  CRIMAC_REQUIRE(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, "wce_fwd: label_bytes=%d", label_bytes);
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "wce_fwd: ncls=%d unsupported (2..4)", ncls);
  const long HW = (long)H * W, npix = B * HW;
  // every workgroup ends with two fp64 atomics on the SAME two addresses, which the memory serialises: few, long workgroups
  int grid = grid_for(npix, 256 * 8);
  if (grid > 512) grid = 512;
#define WF(NC)                                                                                     \
  hipLaunchKernelGGL(wce_fwd_kernel<NC>, dim3(grid), dim3(256), 0, ST, logits, labels, label_bytes, \
                     class_w, ignore_index, npix, HW, sums)
  if (ncls == 2) WF(2); else if (ncls == 3) WF(3); else WF(4);
#undef WF
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_wce_bwd(const float* logits, const void* labels, int label_bytes,
                              const float* class_w, int ncls, int ignore_index, int B, int H, int W,
                              const double* sums, float upstream, float* dlogits, void* stream) {
  CRIMAC_REQUIRE(logits && labels && class_w && sums && dlogits && B > 0 && H > 0 && W > 0,
                 "wce_bwd: bad arguments");
  CRIMAC_REQUIRE(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, "wce_bwd: label_bytes=%d", label_bytes);
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "wce_bwd: ncls=%d unsupported (2..4)", ncls);
  const long HW = (long)H * W, npix = B * HW;
  const int grid = grid_for(npix, 256 * 4);
#define WB(NC)                                                                                     \
  hipLaunchKernelGGL(wce_bwd_kernel<NC>, dim3(whole_rounds(wce_bwd_kernel<NC>, 0, grid)), dim3(256), 0, ST, logits, labels, label_bytes, \
                     class_w, ignore_index, npix, HW, sums, upstream, dlogits)
  if (ncls == 2) WB(2); else if (ncls == 3) WB(3); else WB(4);
#undef WB
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_softmax_nchw(const float* logits, float* out, int B, int ncls, int H, int W, void* stream) {
  CRIMAC_REQUIRE(logits && out && B > 0 && H > 0 && W > 0, "softmax_nchw: bad arguments");
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "softmax_nchw: ncls=%d unsupported (2..4)", ncls);
  const long HW = (long)H * W, npix = B * HW;
  const int grid = grid_for(npix, 256 * 4);
#define SM(NC) hipLaunchKernelGGL(softmax_nchw_kernel<NC>, dim3(whole_rounds(softmax_nchw_kernel<NC>, 0, grid)), dim3(256), 0, ST, logits, out, npix, HW)
  if (ncls == 2) SM(2); else if (ncls == 3) SM(3); else SM(4);
#undef SM
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_sgd_momentum(float* p, float* g, float* v, long n, float lr, float momentum,
                                   float grad_scale, int zero_grad, void* stream) {
  CRIMAC_REQUIRE(p && g && v && n > 0, "sgd_momentum: bad arguments");
  CRIMAC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)v % 16 == 0),
                 "sgd_momentum: buffers must be 16-byte aligned");
  const long n4 = n / 4;
  const int grid = grid_for(n4 > 0 ? n4 : 1, 256 * 4);
  hipLaunchKernelGGL(sgd_kernel, dim3(whole_rounds(sgd_kernel, 0, grid)), dim3(256), 0, ST, p, g, v, n4, n, lr, momentum, grad_scale,
                     zero_grad, (int*)nullptr);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_grad_overflow_flag(const float* g, long n, int* state, void* stream) {
  CRIMAC_REQUIRE(g && state && n > 0 && ((uintptr_t)g % 16 == 0), "grad_overflow_flag: bad arguments");
  const long n4 = n / 4;
  hipLaunchKernelGGL(grad_overflow_kernel, dim3(grid_for(n4 > 0 ? n4 : 1, 256 * 8)), dim3(256), 0, ST, g, n4, n, state);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_sgd_momentum_guarded(float* p, float* g, float* v, long n, float lr, float momentum,
                                           float grad_scale, int zero_grad, int* state, void* stream) {
  CRIMAC_REQUIRE(p && g && v && state && n > 0, "sgd_momentum_guarded: bad arguments");
  CRIMAC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)v % 16 == 0),
                 "sgd_momentum_guarded: buffers must be 16-byte aligned");
  const long n4 = n / 4;
  const int grid = grid_for(n4 > 0 ? n4 : 1, 256 * 4);
  hipLaunchKernelGGL(sgd_kernel, dim3(whole_rounds(sgd_kernel, 0, grid)), dim3(256), 0, ST, p, g, v, n4, n, lr, momentum, grad_scale,
                     zero_grad, state);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
