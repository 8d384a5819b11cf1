// Late metadata injection (reference: UNet_LateMetInject, crimac_unet/models/unet.py:346-391, and
// MetaPostProcessing, unet.py:140-166) for CDNA4 (gfx950).
//
//   m       = W3 . relu(W2 . relu(W1 . meta + b1) + b2) + b3          per pixel, meta [B][Cm][H][W] fp32 (NCHW, as
//                                                                    the reference Dataset hands it over)
//   logits  = conv_final([x | m])  =  Wf[:, :64] . x  +  Wf[:, 64] * m  +  bf         (torch.cat((x, m), 1), :386-388)
//
// The 64-channel part of the head is crimac_head_fwd / crimac_head_bwd (elementwise.hip) on the first 64 columns of
// conv_final.weight; the kernels here add the metadata column and its 3-layer perceptron.  All of it is HBM-bound
// fp32 work on [B*H*W] pixels (Cm <= 8 input planes, one output plane): one thread per pixel, the 1.2 K perceptron
// weights in LDS (broadcast reads), hidden activations in registers.  The weight gradients of the 32 x 32 layer are an
// outer-product sum over pixels: every 256-pixel block stages (dh2, h1) in LDS and each thread accumulates four of
// the 1024 products over the block's pixels; workgroup partials go out by fp32 atomics once per workgroup.
#include "common.h"

namespace {

constexpr int HID = 32;          // MetaPostProcessing.hidden_channels_1/2 (unet.py:147-148)
constexpr int MAX_CM = 8;        // get_in_channels (pipeline.py:413-425) yields at most 7

struct MlpW {
  const float *w1, *b1, *w2, *b2, *w3, *b3;      // [32][Cm], [32], [32][32], [32], [1][32], [1]
};

__device__ __forceinline__ void load_weights(const MlpW& w, int Cm, float* s) {
  // LDS image: w1 [32][MAX_CM] | b1 [32] | w2 [32][32] | b2 [32] | w3 [32] | b3 [1]
  for (int i = threadIdx.x; i < HID * MAX_CM; i += blockDim.x) {
    const int j = i / MAX_CM, c = i % MAX_CM;
    s[i] = c < Cm ? w.w1[j * Cm + c] : 0.f;
  }
  float* p = s + HID * MAX_CM;
  for (int i = threadIdx.x; i < HID; i += blockDim.x) p[i] = w.b1[i];
  p += HID;
  for (int i = threadIdx.x; i < HID * HID; i += blockDim.x) p[i] = w.w2[i];
  p += HID * HID;
  for (int i = threadIdx.x; i < HID; i += blockDim.x) p[i] = w.b2[i];
  p += HID;
  for (int i = threadIdx.x; i < HID; i += blockDim.x) p[i] = w.w3[i];
  p += HID;
  if (threadIdx.x == 0) p[0] = w.b3[0];
}
constexpr int W_FLOATS = HID * MAX_CM + HID + HID * HID + HID + HID + 1;

__device__ __forceinline__ float mlp_forward(const float* s, const float (&x)[MAX_CM], float (&h1)[HID], float (&h2)[HID]) {
  const float *w1 = s, *b1 = s + HID * MAX_CM, *w2 = b1 + HID, *b2 = w2 + HID * HID, *w3 = b2 + HID, *b3 = w3 + HID;
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    float a = b1[j];
#pragma unroll
    for (int c = 0; c < MAX_CM; ++c) a += w1[j * MAX_CM + c] * x[c];
    h1[j] = fmaxf(a, 0.f);
  }
  float m = b3[0];
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    float a = b2[j];
#pragma unroll
    for (int i = 0; i < HID; ++i) a += w2[j * HID + i] * h1[i];
    h2[j] = fmaxf(a, 0.f);
    m += w3[j] * h2[j];
  }
  return m;
}

__global__ __launch_bounds__(256) void meta_mlp_fwd_kernel(const float* __restrict__ meta, int Cm, long npix, long HW,
                                                           MlpW w, float* __restrict__ m_out) {
  __shared__ float s[W_FLOATS];
  load_weights(w, Cm, s);
  __syncthreads();
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW, hw = p - b * HW;
    float x[MAX_CM], h1[HID], h2[HID];
#pragma unroll
    for (int c = 0; c < MAX_CM; ++c) x[c] = c < Cm ? meta[(b * Cm + c) * HW + hw] : 0.f;
    m_out[p] = mlp_forward(s, x, h1, h2);
  }
}

// logits[b][o][hw] += wm[o] * m[b*HW + hw]   (+ softmax over the classes)
template <int NC>
__global__ __launch_bounds__(256) void meta_inject_fwd_kernel(const float* __restrict__ m, const float* __restrict__ wm,
                                                              float* __restrict__ logits, long npix, long HW,
                                                              int softmax) {
  float wv[NC];
#pragma unroll
  for (int o = 0; o < NC; ++o) wv[o] = wm[o];
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW, hw = p - b * HW;
    const float mv = m[p];
    float z[NC];
#pragma unroll
    for (int o = 0; o < NC; ++o) z[o] = logits[(b * NC + o) * HW + hw] + wv[o] * mv;
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
#pragma unroll
    for (int o = 0; o < NC; ++o) logits[(b * NC + o) * HW + hw] = z[o];
  }
}

// Backward of the metadata column and of the perceptron:
//   dwm[o] += sum_p dl[o][p] * m[p];   dm[p] = sum_o dl[o][p] * wm[o];   then the three Linear layers.
template <int NC>
__global__ __launch_bounds__(256) void meta_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ meta, int Cm,
                                                       long npix, long HW, const float* __restrict__ wm, MlpW w,
                                                       float* dwm, float* gw1, float* gb1, float* gw2, float* gb2,
                                                       float* gw3, float* gb3) {
  // dynamic LDS (116 KB): weights | h1 | dh2 | dh1 of the block's pixels (row pitch 33: conflict-free column walks) |
  // metadata inputs | reduction scratch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  float (*sh1)[HID + 1] = reinterpret_cast<float (*)[HID + 1]>(s + ((W_FLOATS + 3) & ~3));
  float (*sd2)[HID + 1] = sh1 + 256;
  float (*sd1)[HID + 1] = sd2 + 256;
  float (*sx)[MAX_CM + 1] = reinterpret_cast<float (*)[MAX_CM + 1]>(sd1 + 256);
  float* red = reinterpret_cast<float*>(sx + 256);
  load_weights(w, Cm, s);
  const float *w2 = s + HID * MAX_CM + HID, *w3 = w2 + HID * HID + HID;
  float wv[NC];
#pragma unroll
  for (int o = 0; o < NC; ++o) wv[o] = wm[o];
  const int t = threadIdx.x;
  // this thread's four entries of dW2 [j][i0..i0+3], its entries of dW1 [j1][c] (t < 32 * MAX_CM), and of the vectors
  const int j2 = t >> 3, i0 = (t & 7) * 4;
  float a2[4] = {0.f, 0.f, 0.f, 0.f}, a1 = 0.f, av = 0.f;       // av: gb1/gb2/gw3 by thread role (below)
  const int j1 = t / MAX_CM, c1 = t % MAX_CM;
  float dwm_acc[NC], gb3_acc = 0.f;
#pragma unroll
  for (int o = 0; o < NC; ++o) dwm_acc[o] = 0.f;
  __syncthreads();
  for (long p0 = (long)blockIdx.x * 256; p0 < npix; p0 += (long)gridDim.x * 256) {
    const long p = p0 + t;
    const bool ok = p < npix;
    float x[MAX_CM], h1[HID], h2[HID];
    float dm = 0.f, mval = 0.f;
    if (ok) {
      const long b = p / HW, hw = p - b * HW;
#pragma unroll
      for (int c = 0; c < MAX_CM; ++c) x[c] = c < Cm ? meta[(b * Cm + c) * HW + hw] : 0.f;
      mval = mlp_forward(s, x, h1, h2);
#pragma unroll
      for (int o = 0; o < NC; ++o) {
        const float g = dl[(b * NC + o) * HW + hw];
        dm += g * wv[o];
        dwm_acc[o] += g * mval;
      }
    } else {
#pragma unroll
      for (int c = 0; c < MAX_CM; ++c) x[c] = 0.f;
#pragma unroll
      for (int j = 0; j < HID; ++j) { h1[j] = 0.f; h2[j] = 0.f; }
    }
    gb3_acc += dm;
    float dh2[HID], dh1[HID];
#pragma unroll
    for (int i = 0; i < HID; ++i) dh1[i] = 0.f;
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      dh2[j] = h2[j] > 0.f ? dm * w3[j] : 0.f;
#pragma unroll
      for (int i = 0; i < HID; ++i) dh1[i] += w2[j * HID + i] * dh2[j];
    }
#pragma unroll
    for (int i = 0; i < HID; ++i) {
      dh1[i] = h1[i] > 0.f ? dh1[i] : 0.f;
      sh1[t][i] = h1[i];
      sd1[t][i] = dh1[i];
      sd2[t][i] = dh2[i];
    }
#pragma unroll
    for (int c = 0; c < MAX_CM; ++c) sx[t][c] = x[c];
    // gw3[j] += dm * h2[j]: fold over the block through the dh2 image later would lose h2; do it by wave shuffles
    // of the 32 values is expensive -- instead reuse sd1 row-major AFTER the outer products (below).
    __syncthreads();
    // outer products over the block's 256 pixels
    for (int q = 0; q < 256; ++q) {
      const float d2 = sd2[q][j2];
#pragma unroll
      for (int k = 0; k < 4; ++k) a2[k] += d2 * sh1[q][i0 + k];
    }
    if (t < HID * MAX_CM) {
      for (int q = 0; q < 256; ++q) a1 += sd1[q][j1] * sx[q][c1];
    }
    // vectors: threads 0..31 -> gb2[j] = sum dh2[j]; 32..63 -> gb1[j] = sum dh1[j]
    if (t < HID) {
      for (int q = 0; q < 256; ++q) av += sd2[q][t];
    } else if (t < 2 * HID) {
      for (int q = 0; q < 256; ++q) av += sd1[q][t - HID];
    }
    __syncthreads();
    // gw3[j] += dm * h2[j]: stage (dm * h2) in sd2 and let threads 64..95 sum the columns
#pragma unroll
    for (int j = 0; j < HID; ++j) sd2[t][j] = dm * h2[j];
    __syncthreads();
    if (t >= 2 * HID && t < 3 * HID) {
      for (int q = 0; q < 256; ++q) av += sd2[q][t - 2 * HID];
    }
    __syncthreads();
  }
  // workgroup partials -> global
#pragma unroll
  for (int k = 0; k < 4; ++k) atomicAdd(&gw2[j2 * HID + i0 + k], a2[k]);
  if (t < HID * MAX_CM && c1 < Cm) atomicAdd(&gw1[j1 * Cm + c1], a1);
  if (t < HID) atomicAdd(&gb2[t], av);
  else if (t < 2 * HID) atomicAdd(&gb1[t - HID], av);
  else if (t < 3 * HID) atomicAdd(&gw3[t - 2 * HID], av);
  // dwm, gb3: wave reduce, then LDS, then one atomic each
  float v[NC + 1];
#pragma unroll
  for (int o = 0; o < NC; ++o) v[o] = wave_sum(dwm_acc[o]);
  v[NC] = wave_sum(gb3_acc);
  if ((t & 63) == 0)
    for (int o = 0; o <= NC; ++o) red[(t >> 6) * 8 + o] = v[o];
  __syncthreads();
  if (t <= NC) {
    const float sum = red[t] + red[8 + t] + red[16 + t] + red[24 + t];
    if (t < NC) atomicAdd(&dwm[t], sum);
    else atomicAdd(&gb3[0], sum);
  }
}


// ---- metadata planes of a crop (batch/dataset.py:288-351, get_crop_memmap) -----------------------------------------------
// Seven planes at most, every one a function of the crop centre and of three per-ping vectors of the echogram:
//   portion_year      : the echogram's scalar
//   portion_day (x2)  : sin / cos of 2 pi * portion_of_day_vector[centre ping]           (index clamped: < 0 -> 0, >= n -> last)
//   time_diff         : time_vector_diff[ping of the column]                             (same clamping, per column)
//   depth_rel         : row / seabed[ping]         depth_abs_surface : row / H         depth_abs_seabed : (seabed[ping] - row) / H
// with row = cy - H/2 + y, ping = cx - W/2 + x -- the reference's arange(c - w // 2, c + w // 2), one pixel up / left of the
// DATA crop's grid (getGrid: c - (w + 1) // 2 + 1 ...): reproduced, not "fixed".  The reference computes in float64 and the
// batch is cast to float32 by SegPipe.predict_batch (.float()): the same here (double arithmetic, one rounding).
struct MetaPlaneArgs {
  const int* centres;            // [P][2] (range idx, ping idx)
  int P, H, W, flags;            // flags: bit 0 portion_year, 1 portion_day, 2 time_diff, 3 depth_rel, 4 depth_abs_surface, 5 depth_abs_seabed
  double portion_year;
  const double* portion_day; int n_day;
  const double* time_diff; int n_td;
  const long long* seabed; int n_sb;
  float* out;                    // [P][Cm][H][W]
  int Cm;
};
__device__ __forceinline__ int clamp_last(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

__global__ __launch_bounds__(256) void meta_planes_kernel(MetaPlaneArgs a) {
  const long HW = (long)a.H * a.W, total = (long)a.P * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i / HW);
    const int rem = (int)(i - p * HW);
    const int y = rem / a.W, x = rem - y * a.W;
    const int cy = a.centres[2 * p], cx = a.centres[2 * p + 1];
    const int row = cy - a.H / 2 + y, ping = cx - a.W / 2 + x;
    float* o = a.out + ((long)p * a.Cm) * HW + rem;
    int c = 0;
    if (a.flags & 1) { o[c * HW] = (float)a.portion_year; ++c; }
    if (a.flags & 2) {
      const double t = a.portion_day[clamp_last(cx, a.n_day)];
      o[c * HW] = (float)sin(2.0 * 3.141592653589793 * t); ++c;
      o[c * HW] = (float)cos(2.0 * 3.141592653589793 * t); ++c;
    }
    if (a.flags & 4) { o[c * HW] = (float)a.time_diff[clamp_last(ping, a.n_td)]; ++c; }
    if (a.flags & 56) {
      const double sb = (double)a.seabed[clamp_last(ping, a.n_sb)];
      if (a.flags & 8) { o[c * HW] = (float)((double)row / sb); ++c; }
      if (a.flags & 16) { o[c * HW] = (float)((double)row / (double)a.H); ++c; }
      if (a.flags & 32) { o[c * HW] = (float)((sb - (double)row) / (double)a.H); ++c; }
    }
  }
}
}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int crimac_meta_mlp_fwd(const float* meta, int Cm, int B, int H, int W, const float* w1, const float* b1,
                                   const float* w2, const float* b2, const float* w3, const float* b3, float* m,
                                   void* stream) {
  CRIMAC_REQUIRE(meta && w1 && b1 && w2 && b2 && w3 && b3 && m && B > 0 && H > 0 && W > 0, "meta_mlp_fwd: bad arguments");
  CRIMAC_REQUIRE(Cm >= 1 && Cm <= MAX_CM, "meta_mlp_fwd: %d metadata channels (1..%d supported)", Cm, MAX_CM);
  const long HW = (long)H * W, npix = B * HW;
  long blocks = (npix + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(meta_mlp_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, meta, Cm, npix, HW,
                     MlpW{w1, b1, w2, b2, w3, b3}, m);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_meta_inject_fwd(const float* m, const float* wm, float* logits, int B, int H, int W, int ncls,
                                      int softmax, void* stream) {
  CRIMAC_REQUIRE(m && wm && logits && B > 0 && H > 0 && W > 0, "meta_inject_fwd: bad arguments");
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "meta_inject_fwd: ncls=%d unsupported (2..4)", ncls);
  const long HW = (long)H * W, npix = B * HW;
  long blocks = (npix + 255) / 256;
  if (blocks > 2048) blocks = 2048;
#define MI(NC) hipLaunchKernelGGL(meta_inject_fwd_kernel<NC>, dim3((unsigned)blocks), dim3(256), 0, ST, m, wm, logits, npix, HW, softmax)
  if (ncls == 2) MI(2); else if (ncls == 3) MI(3); else MI(4);
#undef MI
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_meta_bwd(const float* dlogits, const float* meta, int Cm, int B, int H, int W, int ncls,
                               const float* wm, const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* w3, const float* b3, float* dwm, float* gw1, float* gb1, float* gw2,
                               float* gb2, float* gw3, float* gb3, void* stream) {
  CRIMAC_REQUIRE(dlogits && meta && wm && w1 && b1 && w2 && b2 && w3 && b3 && dwm && gw1 && gb1 && gw2 && gb2 && gw3 &&
                     gb3 && B > 0 && H > 0 && W > 0,
                 "meta_bwd: bad arguments");
  CRIMAC_REQUIRE(Cm >= 1 && Cm <= MAX_CM, "meta_bwd: %d metadata channels (1..%d supported)", Cm, MAX_CM);
  CRIMAC_REQUIRE(ncls >= 2 && ncls <= 4, "meta_bwd: ncls=%d unsupported (2..4)", ncls);
  const long HW = (long)H * W, npix = B * HW;
  long blocks = (npix + 255) / 256;
  if (blocks > 512) blocks = 512;
  const MlpW w{w1, b1, w2, b2, w3, b3};
  const size_t lds = sizeof(float) * (((W_FLOATS + 3) & ~3) + 3 * 256 * (HID + 1) + 256 * (MAX_CM + 1) + 64);
  static unsigned long long attr_devs = 0;
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&meta_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&meta_bwd_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&meta_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
#define MB(NC) hipLaunchKernelGGL(meta_bwd_kernel<NC>, dim3((unsigned)blocks), dim3(256), lds, ST, dlogits, meta, Cm, npix, HW, wm, w, dwm, gw1, gb1, gw2, gb2, gw3, gb3)
  if (ncls == 2) MB(2); else if (ncls == 3) MB(3); else MB(4);
#undef MB
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}


// Metadata planes of P crops (reference batch/dataset.py:288-351, the `meta` half of get_crop_memmap's result): what the
// reference's Dataset builds per patch in numpy DataLoader workers, built on the GPU from the echogram's three per-ping
// vectors.  flags: bit 0 portion_year, 1 portion_day (two planes: sin, cos), 2 time_diff, 3 depth_rel,
// 4 depth_abs_surface, 5 depth_abs_seabed; the planes come out in that order, out [P][Cm][H][W] fp32.
extern "C" int crimac_meta_planes(const int* centres, int P, int H, int W, int flags, double portion_year,
                                  const double* portion_day, int n_day, const double* time_diff, int n_td,
                                  const long long* seabed, int n_sb, float* out, void* stream) {
  CRIMAC_REQUIRE(centres && out && P > 0 && H > 0 && W > 0 && flags > 0 && flags < 64, "meta_planes: bad arguments");
  CRIMAC_REQUIRE(!(flags & 2) || (portion_day && n_day > 0), "meta_planes: portion_day needs its vector");
  CRIMAC_REQUIRE(!(flags & 4) || (time_diff && n_td > 0), "meta_planes: time_diff needs its vector");
  CRIMAC_REQUIRE(!(flags & 56) || (seabed && n_sb > 0), "meta_planes: the depth planes need the seabed vector");
  MetaPlaneArgs a{centres, P, H, W, flags, portion_year, portion_day, n_day, time_diff, n_td, seabed, n_sb, out, 0};
  a.Cm = (flags & 1) + 2 * ((flags >> 1) & 1) + ((flags >> 2) & 1) + ((flags >> 3) & 1) + ((flags >> 4) & 1) + ((flags >> 5) & 1);
  const long total = (long)P * H * W;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(meta_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
