// On-GPU training augmentation fused with the data transform (CDNA4 / gfx950), HBM-bound.
//
// Reference (numpy, in DataLoader workers): add_noise (batch/data_augmentation/add_noise.py:21-41) then
// flip_x_axis (flip_x_axis.py:21-25), composed by define_data_augmentation (batch/transforms.py:39-46),
// followed by remove_nan_inf + db_with_limits (remove_nan_inf.py:23-34, db_with_limits.py:20-24):
//   with p = .5 per sample: 5 % of the (channel, pixel) values are multiplied by U(1,10) (half of
//   them) or U(0,1) (the other half); with p = .5 per sample the ping axis of data and labels is
//   flipped; non-finite -> 0 (label -> -100 where channel 0 is non-finite); 10*log10(x+1e-10) in [-75,0].
// One pass: NCHW fp32 linear sv in -> NHWC activations (the first conv's input) + int16 labels out.
// Randomness: Philox4x32-10 keyed on (seed, sample), counter = element index -- reproducible and
// independent of the launch geometry; oracle/augment_oracle.py restates the same generator in numpy.
#include "common.h"

namespace {

struct u4 { unsigned x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    u4 n;
    n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0;
    n.y = (unsigned)p1;
    n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1;
    n.w = (unsigned)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

template <typename T>
__global__ __launch_bounds__(256) void augment_db_kernel(
    const float* __restrict__ data, const void* __restrict__ labels_in, int label_bytes,
    T* __restrict__ out, short* __restrict__ labels_out, unsigned char* __restrict__ aux, int thr_channel,
    float thr_lo, float thr_hi, int B, int C, int H, int W, int ld,
    unsigned seed_lo, unsigned seed_hi, int do_noise, int do_flip, int db_scaled, float p_apply, float p_change) {
  const long HW = (long)H * W, npix = (long)B * HW;
  for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix;
       pix += (long)gridDim.x * blockDim.x) {
    const int b = (int)(pix / HW);
    const long hw = pix % HW;
    const int y = (int)(hw / W), x = (int)(hw % W);
    // per-sample decisions: counter (0,0,0,0xA5A5A5A5), key (seed_lo ^ b, seed_hi)
    const u4 s = philox4x32_10(u4{0u, 0u, 0u, 0xA5A5A5A5u}, seed_lo ^ (unsigned)b, seed_hi);
    const bool noisy = do_noise && u01(s.x) < p_apply;
    const bool flip = do_flip && u01(s.y) < p_apply;
    const int xo = flip ? W - 1 - x : x;
    float v[16];
    bool nonfinite0 = false, thr_hit = false;
    for (int c = 0; c < C; ++c) {
      float d = data[((long)b * C + c) * HW + hw];
      if (noisy) {
        const u4 r = philox4x32_10(u4{(unsigned)(c * HW + hw), (unsigned)((c * HW + hw) >> 32), 1u, 0u},
                                   seed_lo ^ (unsigned)b, seed_hi);
        if (u01(r.x) < p_change) d *= (u01(r.y) < 0.5f) ? (1.0f + 9.0f * u01(r.z)) : u01(r.w);
      }
      if (c == thr_channel) thr_hit = d > thr_lo && d < thr_hi;      // on the augmented linear value; NaN -> false
      if (!isfinite(d)) { if (c == 0) nonfinite0 = true; d = 0.f; }
      d = 10.f * log10f(d + 1e-10f);
      d = fminf(fmaxf(d, -75.f), 0.f);
      // db_with_limits_scaled (db_with_limits.py:27-33), the data transform of the metadata configurations
      // (batch/transforms.py:50-51): 1 + dB / |limit_low| in [0, 1]
      v[c] = db_scaled ? 1.f + d / 75.f : d;
    }
    for (int c = C; c < ld; ++c) v[c] = 0.f;
    T* dst = out + (((long)b * H + y) * W + xo) * ld;
    for (int c0 = 0; c0 < ld; c0 += 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = v[c0 + j];
      store8(dst + c0, t);
    }
    if (labels_out) {
      long l = 0;
      if (labels_in) {
        if (label_bytes == 8) l = ((const long long*)labels_in)[pix];
        else if (label_bytes == 4) l = ((const int*)labels_in)[pix];
        else l = ((const short*)labels_in)[pix];
      }
      // with `aux` the label transform still has to run (crimac_refine_labels): hand over the raw ids and
      // the two per-pixel facts it needs from the data, the NaN rule is applied there, after it
      if (nonfinite0 && !aux) l = -100;
      labels_out[((long)b * H + y) * W + xo] = (short)l;
    }
    if (aux) aux[((long)b * H + y) * W + xo] = (unsigned char)((thr_hit ? 1 : 0) | (nonfinite0 ? 2 : 0));
  }
}

// flip_x_axis_metadata (flip_x_axis.py:27-32) for the planes that do not go through augment_db_kernel -- the metadata
// planes of UNet_LateMetInject: the ping axis of sample b is flipped under the SAME per-sample decision (same Philox
// draw) as its data and labels.
__global__ __launch_bounds__(256) void flip_planes_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                          int C, int H, int W, unsigned seed_lo, unsigned seed_hi,
                                                          int do_flip, float p_apply) {
  const long CHW = (long)C * H * W, n = (long)B * CHW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / CHW);
    const u4 s = philox4x32_10(u4{0u, 0u, 0u, 0xA5A5A5A5u}, seed_lo ^ (unsigned)b, seed_hi);
    const bool flip = do_flip && u01(s.y) < p_apply;
    const int x = (int)(i % W);
    out[flip ? i - x + (W - 1 - x) : i] = in[i];
  }
}

}  // namespace

extern "C" int crimac_augment_flip_planes(const float* in, float* out, int B, int C, int H, int W,
                                          unsigned long long seed, int do_flip, void* stream) {
  CRIMAC_REQUIRE(in && out && in != out && B > 0 && C > 0 && H > 0 && W > 0, "augment_flip_planes: bad arguments");
  const long n = (long)B * C * H * W;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flip_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, B, C, H, W,
                     (unsigned)seed, (unsigned)(seed >> 32), do_flip, 0.5f);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_augment_db_nhwc(int prec, const float* data, const void* labels_in, int label_bytes,
                                      void* out, short* labels_out, unsigned char* aux_mask, int thr_channel,
                                      float thr_lo, float thr_hi, int B, int C, int H, int W, long ld,
                                      unsigned long long seed, int do_noise, int do_flip, int db_scaled, void* stream) {
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, "augment_db_nhwc: bad precision %d", prec);
  CRIMAC_REQUIRE(data && out && B > 0 && C > 0 && C <= 16 && H > 0 && W > 0 && ld >= C && ld <= 16 && ld % 8 == 0,
                 "augment_db_nhwc: bad arguments (C=%d ld=%ld)", C, ld);
  CRIMAC_REQUIRE(!labels_in || label_bytes == 2 || label_bytes == 4 || label_bytes == 8,
                 "augment_db_nhwc: label_bytes=%d", label_bytes);
  CRIMAC_REQUIRE(!aux_mask || (thr_channel >= 0 && thr_channel < C), "augment_db_nhwc: bad threshold channel %d",
                 thr_channel);
  if (!aux_mask) thr_channel = -1;
  const long npix = (long)B * H * W;
  long blocks = (npix + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
  hipStream_t st = (hipStream_t)stream;
  CRIMAC_FOR_STORAGE2(prec, TF_, T, hipLaunchKernelGGL(augment_db_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, data,
                                                 labels_in, label_bytes, (T*)out, labels_out, aux_mask, thr_channel,
                                                 thr_lo, thr_hi, B, C, H, W, (int)ld, lo, hi, do_noise, do_flip, db_scaled, 0.5f,
                                                 0.05f));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
