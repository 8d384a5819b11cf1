// Tiled whole-survey inference plumbing on the GPU (CDNA4 / gfx950), HBM-bound byte movers.
//
// Reference: save_survey_predictions_zarr's per-patch host work (save_predict.py:160-209):
//   * DatasetGriddedReader.get_preload_data_labels -> new_get_crop_3d (dataset.py:192-205,
//     utils/np.py:361-375): gather a 256x256 crop around each grid centre from the preloaded chunk,
//     0 outside the chunk;
//   * remove_nan_inf + db_with_limits (remove_nan_inf.py:23-34, db_with_limits.py:20-24, :36-38):
//     non-finite -> 0, 10*log10(x + 1e-10) clamped to [-75, 0];
//   * fill_out_array (save_predict.py:41-65) with the validity rules of the test-time label
//     transforms (convert_label_indexing_unused_species, mask_label_seabed, mask_label_overlap,
//     remove_nan_inf): scatter softmax channels [SANDEEL, OTHER] of each patch's valid interior into
//     the chunk's [2, range, pings] output.
//
// The chunk stays resident in HBM in the reader's own (zarr) orientation [freq][ping][range] (range
// contiguous); the gather transposes through LDS so both the reads (along range) and the NHWC writes
// (along ping, 16 channels per pixel) are coalesced, and the dB transform, channel padding and
// bf16/fp32 conversion are fused into it -- the patch lands directly in the first conv's input layout.
#include "common.h"

namespace {

constexpr int TS = 32;   // tile side

// data [C][Wd][H] fp32; centres [P][2] = (cy, cx_local) with cx_local relative to the chunk slice.
// border_labels != NULL (memm flavour, save_predict.py:222-265): labels [Wd][H] int16 raw annotation ids covering the
// same extent as `data`; a pixel outside that extent, or whose raw label the test-time label transform maps to
// "ignore" (convert_label_indexing: negative ids), gets 0.0 AFTER the dB transform in every channel
// (set_data_border_value, batch/data_transforms/set_data_border_value.py:20-23, last step of define_data_transform_test).
template <typename T>
__global__ __launch_bounds__(256) void gather_patches_kernel(const float* __restrict__ data, int C, int Wd,
                                                             int H, const int* __restrict__ centres,
                                                             int ph, int pw, T* __restrict__ out,
                                                             int ld, const short* __restrict__ border_labels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);      // [C][TS (x)][TS + 1 (y)]
  const int p = blockIdx.z;
  const int ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
  const int cy = centres[2 * p], cx = centres[2 * p + 1];
  const int y_base = cy - ((ph + 1) / 2) + 1 + ty0;        // data row of tile row 0 (np.py:40-46)
  const int x_base = cx - ((pw + 1) / 2) + 1 + tx0;
  const int tx = threadIdx.x & 31, tr = threadIdx.x >> 5;  // tr 0..7
  // read phase: lanes run along range (contiguous in the chunk)
  bool border[4] = {false, false, false, false};
  if (border_labels) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x_base + tr + 8 * k, y = y_base + tx;
      const bool inside = x >= 0 && x < Wd && y >= 0 && y < H;
      border[k] = !inside || border_labels[(long)x * H + y] < 0;
    }
  }
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xi = tr + 8 * k;                 // tile column (ping)
      const int x = x_base + xi, y = y_base + tx;
      float v = 0.f;                             // boundary_val_data = 0 (dataset.py:195)
      if (x >= 0 && x < Wd && y >= 0 && y < H && (ty0 + tx) < ph && (tx0 + xi) < pw)
        v = data[((long)c * Wd + x) * H + y];
      if (!isfinite(v)) v = 0.f;                 // remove_nan_inf
      v = 10.f * log10f(v + 1e-10f);             // db_with_limits
      v = fminf(fmaxf(v, -75.f), 0.f);
      if (border[k]) v = 0.f;                    // set_data_border_value
      tile[(c * TS + xi) * (TS + 1) + tx] = v;
    }
  }
  __syncthreads();
  // write phase: lanes run along ping (contiguous pixels of the NHWC patch)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yi = tr + 8 * k;
    const int py = ty0 + yi, px = tx0 + tx;
    if (py >= ph || px >= pw) continue;
    T* dst = out + (((long)p * ph + py) * pw + px) * ld;
    for (int c0 = 0; c0 < ld; c0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c0 + j) < C ? tile[((c0 + j) * TS + tx) * (TS + 1) + yi] : 0.f;
      store8(dst + c0, v);
    }
  }
}

// probs [P][ncls][ph][pw] fp32; centres [P][2] global (cy, cx); out [2][H][n_chunk] fp32 (or fp16).
struct ScatterParams {
  const float* probs; int ncls; const int* centres; int P, ph, pw, overlap, start_ping, n_chunk, H;
  const short* labels;
  const unsigned char* seabed_mask; int mask_ping0, mask_pings;
  const int* seabed; int seabed_ping0, seabed_pings;     // seabed index per ping: replaces seabed_mask when given
  const float* data0; int data_ping0, data_pings;
  int seabed_pad;
  int seabed_rule;          // 0: zarr reader (pad shifts the mask inside the patch's slice), 1: Echogram (absolute rows)
  void* out; int out_f16;
};
__global__ __launch_bounds__(256) void scatter_patches_kernel(ScatterParams q) {
  const int ph = q.ph, pw = q.pw, overlap = q.overlap, H = q.H;
  const int iw = pw - 2 * overlap, ih = ph - 2 * overlap;
  const long per_patch = (long)ih * iw;
  const long total = per_patch * q.P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i / per_patch);
    const int r = (int)(i % per_patch);
    const int py = overlap + r / iw, px = overlap + r % iw;      // mask_label_overlap: rim excluded
    const int cy = q.centres[2 * p], cx = q.centres[2 * p + 1];
    const int y = cy - ph / 2 + 1 + py, x = cx - pw / 2 + 1 + px;   // patch_coord_to_data_coord
    const int xl = x - q.start_ping;
    if (y < 0 || y >= H || xl < 0 || xl >= q.n_chunk) continue;     // label crop out of range: -100
    int lab = q.labels ? (int)q.labels[(long)xl * H + y] : 0;
    if (lab < 0) continue;                                           // convert_label_indexing: -100
    if (lab == 0 && (q.seabed_mask || q.seabed)) {                   // mask_label_seabed (background only)
      // zarr reader: rows of the patch's own slice, shifted down by the pad inside the slice; Echogram: absolute
      const int y_top = q.seabed_rule == 0 ? max(cy - ph / 2 + 1, 0) : 0;
      if (y - y_top >= q.seabed_pad) {
        bool below = false;
        if (q.seabed) {
          const int xs = x - q.seabed_ping0;
          below = xs >= 0 && xs < q.seabed_pings && (y - q.seabed_pad) >= q.seabed[xs];
        } else {
          const int xm = x - q.mask_ping0;
          below = xm >= 0 && xm < q.mask_pings && q.seabed_mask[(long)xm * H + (y - q.seabed_pad)];
        }
        if (below) continue;
      }
    }
    if (q.data0) {                                                   // remove_nan_inf: ch 0 non-finite
      const int xd = x - q.data_ping0;
      if (xd >= 0 && xd < q.data_pings && !isfinite(q.data0[(long)xd * H + y])) continue;
    }
    const long src = (((long)p * q.ncls + 1) * ph + py) * pw + px;   // channel SANDEEL = 1
    const long d0 = ((long)0 * H + y) * q.n_chunk + xl, d1 = ((long)1 * H + y) * q.n_chunk + xl;
    if (q.out_f16) {                                                 // the reference stores float16 (save_predict.py:212, :252)
      reinterpret_cast<half_t*>(q.out)[d0] = (half_t)q.probs[src];
      reinterpret_cast<half_t*>(q.out)[d1] = (half_t)q.probs[src + (long)ph * pw];
    } else {
      reinterpret_cast<float*>(q.out)[d0] = q.probs[src];
      reinterpret_cast<float*>(q.out)[d1] = q.probs[src + (long)ph * pw];   // channel OTHER = 2
    }
  }
}

// Validation metrics (pipeline.py:242-341): histogram of the float16-rounded SANDEEL probability over
// the valid pixels, split by "label == SANDEEL".  The reference gathers every pixel's probability as
// float16 on the host and sorts them in sklearn's precision_recall_curve; float16 has < 15362
// non-negative values up to 1.0, so the two histograms hold exactly the same information.
__global__ __launch_bounds__(256) void pr_histogram_kernel(const float* __restrict__ logits, int ncls,
                                                           const void* __restrict__ labels, int lbytes,
                                                           long npix, long HW, unsigned int* hist_pos,
                                                           unsigned int* hist_neg) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    long l = lbytes == 8 ? ((const long long*)labels)[p] : lbytes == 4 ? ((const int*)labels)[p]
                                                                       : ((const short*)labels)[p];
    // set_label_ignore_val (pipeline.py:222-239): overlap / refined boundary / boundary / unused -> ignore
    if (l == -70 || l == -30 || l == -100 || l == -10) continue;
    const bool seabed = l == -50;            // below seabed: counts as background with probability 0
    float prob = 0.f;
    if (!seabed) {
      const long b = p / HW, hw = p % HW;
      float z[8], mx = -INFINITY, den = 0.f;
      for (int o = 0; o < ncls; ++o) { z[o] = logits[(b * ncls + o) * HW + hw]; mx = fmaxf(mx, z[o]); }
      for (int o = 0; o < ncls; ++o) { z[o] = expf(z[o] - mx); den += z[o]; }
      prob = z[1] / den;
    }
    const _Float16 h = (_Float16)prob;       // round-to-nearest-even, as numpy .astype(float16)
    unsigned short bits = *reinterpret_cast<const unsigned short*>(&h);
    // a probability is in [0, 1] = bit patterns 0 .. 0x3C00; a NaN (diverged network: 0x7E00 / 0xFE00) must not
    // index past the 16384 bins -> counted in the last bin, which no probability reaches (the host raises on it,
    // as sklearn's precision_recall_curve does on a NaN score)
    if (bits > 0x3C00) bits = CRIMAC_PR_NAN_BIN;
    atomicAdd((!seabed && l == 1) ? &hist_pos[bits] : &hist_neg[bits], 1u);
  }
}

}  // namespace

extern "C" int crimac_pr_histogram(const float* logits, int ncls, const void* labels, int label_bytes,
                                   int B, int H, int W, unsigned int* hist_pos, unsigned int* hist_neg,
                                   void* stream) {
  CRIMAC_REQUIRE(logits && labels && hist_pos && hist_neg && B > 0 && H > 0 && W > 0 && ncls >= 2 && ncls <= 8,
                 "pr_histogram: bad arguments");
  CRIMAC_REQUIRE(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, "pr_histogram: label_bytes=%d", label_bytes);
  const long HW = (long)H * W, npix = B * HW;
  long blocks = (npix + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pr_histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, ncls,
                     labels, label_bytes, npix, HW, hist_pos, hist_neg);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

static int gather_run(int prec, const float* data, int C, int Wd, int H, const int* centres, int P, int ph, int pw,
                      void* out, long ld, const short* border_labels, void* stream) {
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, "gather_patches: bad precision %d", prec);
  CRIMAC_REQUIRE(data && centres && out && C > 0 && C <= 16 && Wd > 0 && H > 0 && P > 0 && ph > 0 && pw > 0,
                 "gather_patches: bad arguments (C=%d must be <= 16)", C);
  CRIMAC_REQUIRE(ld >= C && ld % 8 == 0 && ld <= 16, "gather_patches: ld=%ld must be 8 or 16 and >= C", ld);
  CRIMAC_REQUIRE(P <= 65535, "gather_patches: at most 65535 patches per call");
  dim3 grid((pw + TS - 1) / TS, (ph + TS - 1) / TS, P);
  const size_t lds = (size_t)C * TS * (TS + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  CRIMAC_FOR_STORAGE2(prec, TF_, T, hipLaunchKernelGGL(gather_patches_kernel<T>, grid, dim3(256), lds, st, data, C, Wd, H,
                                                 centres, ph, pw, (T*)out, (int)ld, border_labels));
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_gather_patches(int prec, const float* data, int C, int Wd, int H, const int* centres,
                                     int P, int ph, int pw, void* out, long ld, void* stream) {
  return gather_run(prec, data, C, Wd, H, centres, P, ph, pw, out, ld, nullptr, stream);
}

extern "C" int crimac_gather_patches_memm(int prec, const float* data, int C, int Wd, int H, const int* centres,
                                          int P, int ph, int pw, void* out, long ld, const short* border_labels,
                                          void* stream) {
  CRIMAC_REQUIRE(border_labels, "gather_patches_memm: needs the label array (border rule)");
  return gather_run(prec, data, C, Wd, H, centres, P, ph, pw, out, ld, border_labels, stream);
}

extern "C" int crimac_scatter_patches_ex(const float* probs, int ncls, const int* centres, int P, int ph, int pw,
                                         int overlap, int start_ping, int n_chunk, int H, const short* labels,
                                         const unsigned char* seabed_mask, int mask_ping0, int mask_pings,
                                         const int* seabed, int seabed_ping0, int seabed_pings, const float* data0,
                                         int data_ping0, int data_pings, int seabed_pad, int seabed_rule, void* out,
                                         int out_f16, void* stream) {
  CRIMAC_REQUIRE(probs && centres && out && P > 0 && ph > 0 && pw > 0 && n_chunk > 0 && H > 0,
                 "scatter_patches: bad arguments");
  CRIMAC_REQUIRE(ncls >= 3, "scatter_patches: needs the SANDEEL (1) and OTHER (2) channels, ncls=%d", ncls);
  CRIMAC_REQUIRE(overlap >= 0 && 2 * overlap < ph && 2 * overlap < pw, "scatter_patches: bad overlap %d", overlap);
  CRIMAC_REQUIRE(seabed_rule == 0 || seabed_rule == 1, "scatter_patches: seabed_rule=%d", seabed_rule);
  CRIMAC_REQUIRE(!(seabed_mask && seabed), "scatter_patches: give the seabed mask OR the seabed vector");
  const long total = (long)P * (ph - 2 * overlap) * (pw - 2 * overlap);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ScatterParams q{probs, ncls, centres, P, ph, pw, overlap, start_ping, n_chunk, H, labels, seabed_mask, mask_ping0,
                  mask_pings, seabed, seabed_ping0, seabed_pings, data0, data_ping0, data_pings, seabed_pad,
                  seabed_rule, out, out_f16};
  hipLaunchKernelGGL(scatter_patches_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, q);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_scatter_patches(const float* probs, int ncls, const int* centres, int P, int ph, int pw,
                                      int overlap, int start_ping, int n_chunk, int H, const short* labels,
                                      const unsigned char* seabed_mask, int mask_ping0, int mask_pings,
                                      const float* data0, int data_ping0, int data_pings, int seabed_pad,
                                      float* out, void* stream) {
  return crimac_scatter_patches_ex(probs, ncls, centres, P, ph, pw, overlap, start_ping, n_chunk, H, labels,
                                   seabed_mask, mask_ping0, mask_pings, nullptr, 0, 0, data0, data_ping0, data_pings,
                                   seabed_pad, 0, out, 0, stream);
}
