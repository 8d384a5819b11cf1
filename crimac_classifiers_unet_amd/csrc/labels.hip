// Training label transform on the GPU (CDNA4 / gfx950): one workgroup per 32-row band of a patch, masks in LDS.
//
// Reference (numpy + scipy.ndimage in DataLoader workers), define_label_transform_train
// (batch/transforms.py:71-78), applied to the AUGMENTED linear-sv crop (batch/dataset.py:89-103):
//   refine_label_boundary (batch/label_transforms/refine_label_boundary.py:35-104):
//     crop = bounding box of the pixels whose label is not LABEL_BOUNDARY_VAL (-100); none -> unchanged;
//     m = (label > 0) & (thr_lo < data[thr_channel] < thr_hi);  closed = binary_closing(m[crop], 7x7 disk)
//     (dilation, then erosion, both with 0 outside the CROP);  label > 0 and not closed -> -30;
//   convert_label_indexing (convert_label_indexing.py:24-35): 0 -> 0, 27 -> 1, 1 -> 2, everything else -> -100;
//   remove_nan_inf's label rule (remove_nan_inf.py:30-32, runs after the label transform): -100 where
//   channel 0 is not finite.
// Integer work: bit-exact against oracle/labels_oracle.py, which is pinned on the reference's own output.
#include "common.h"

namespace {

constexpr int kThreads = 1024;

// 7x7 disk (refine_label_boundary.py:50-58): half-width of the row at vertical offset dy
__device__ __forceinline__ int disk_hw(int dy) { return dy == 0 ? 3 : (dy == 1 || dy == -1) ? 3 : (dy == 2 || dy == -2) ? 2 : 1; }

__device__ __forceinline__ long load_label_raw(const void* p, int bytes, long i) {
  if (bytes == 8) return ((const long long*)p)[i];
  if (bytes == 4) return ((const int*)p)[i];
  return ((const short*)p)[i];
}

// Test-time chain (define_label_transform_test, batch/transforms.py:81-99): the labels refine_label_boundary sees
// have already been through convert_label_indexing_unused_species (convert_label_indexing.py:37-47):
// 0 / 27 / 1 -> 0 / 1 / 2, other species (> 0) -> -10, everything else -> -100.
template <bool TEST>
__device__ __forceinline__ long load_label(const void* p, int bytes, long i) {
  const long l = load_label_raw(p, bytes, i);
  if (!TEST) return l;
  return l == 0 ? 0 : l == 27 ? 1 : l == 1 ? 2 : l > 0 ? -10 : -100;
}

// What mask_label_seabed (mask_label_seabed.py:24-68) and mask_label_overlap (mask_label_overlap.py:23-48) need.
struct TestChain {
  const long long* centres;             // [B][2] (range idx, GLOBAL ping idx) of the patch centres
  const int* seabed;                    // per-ping seabed index, pings [seabed_ping0, +seabed_pings) -- or
  int seabed_ping0, seabed_pings;
  const unsigned char* seabed_mask;     // the reader's 2-D mask [mask_pings][n_range] (1 below the seabed)
  int mask_ping0, mask_pings;
  int n_range, seabed_pad, seabed_rule, overlap;
};

// One workgroup per (patch, band of BAND rows): it scans the whole patch for the bounding box (a few reads
// per thread, L2-resident) and then works on its rows plus the 3 + 3 halo rows the two morphology passes need.
constexpr int BAND = 32;

template <bool TEST>
__global__ __launch_bounds__(kThreads) void refine_labels_kernel(
    const void* __restrict__ labels_in, int label_bytes, const unsigned char* __restrict__ aux,
    const float* __restrict__ data, int thr_channel, float thr_lo, float thr_hi, int mode,
    short* __restrict__ labels_out, int C, int H, int W, TestChain tc) {
  extern __shared__ unsigned char smem[];
  const int HW = H * W;
  const int r0 = blockIdx.y * BAND;                     // first output row of this band
  const int rows = min(BAND, H - r0);
  const int m_lo = r0 - 6, m_rows = rows + 12;          // threshold-mask rows held in LDS
  const int d_lo = r0 - 3, d_rows = rows + 6;           // dilation rows held in LDS
  unsigned char* m0 = smem;                             // [m_rows][W]
  unsigned char* dil = smem + (BAND + 12) * W;          // [d_rows][W]
  int* box = reinterpret_cast<int*>(smem + (((2 * BAND + 18) * W + 15) & ~15));   // y0, y1 (exclusive), x0, x1
  const int b = blockIdx.x, tid = threadIdx.x;
  const long base = (long)b * HW;
  if (tid == 0) { box[0] = H; box[1] = 0; box[2] = W; box[3] = 0; }
  __syncthreads();
  // ---- bounding box of label != -100 over the whole patch -------------------------------------------
  int y0 = H, y1 = 0, x0 = W, x1 = 0;
  for (int i = tid; i < HW; i += kThreads) {
    if (load_label<TEST>(labels_in, label_bytes, base + i) != -100) {
      const int y = i / W, x = i - y * W;
      y0 = min(y0, y); y1 = max(y1, y + 1); x0 = min(x0, x); x1 = max(x1, x + 1);
    }
  }
  atomicMin(&box[0], y0); atomicMax(&box[1], y1); atomicMin(&box[2], x0); atomicMax(&box[3], x1);
  __syncthreads();
  y0 = box[0]; y1 = box[1]; x0 = box[2]; x1 = box[3];
  const bool empty = y1 <= y0;
  // ---- threshold mask (0 outside the crop: border_value of both morphology passes) ------------------
  for (int i = tid; i < m_rows * W; i += kThreads) {
    const int ry = i / W, x = i - ry * W, y = m_lo + ry;
    bool hit = false;
    if (!empty && y >= y0 && y < y1 && x >= x0 && x < x1) {
      const long gi = base + (long)y * W + x;
      if (load_label<TEST>(labels_in, label_bytes, gi) > 0) {
        if (aux) hit = aux[gi] & 1;
        else {
          const float d = data[((long)b * C + thr_channel) * HW + (long)y * W + x];
          hit = d > thr_lo && d < thr_hi;
        }
      }
    }
    m0[i] = hit;
  }
  __syncthreads();
  // ---- dilation ------------------------------------------------------------------------------------
  for (int i = tid; i < d_rows * W; i += kThreads) {
    const int ry = i / W, x = i - ry * W, y = d_lo + ry;
    bool v = false;
    if (!empty && y >= y0 && y < y1 && x >= x0 && x < x1) {
      for (int dy = -3; dy <= 3 && !v; ++dy) {
        const int yy = y + dy;
        if (yy < y0 || yy >= y1) continue;
        const int hw = disk_hw(dy);
        const int xa = max(x - hw, x0), xb = min(x + hw, x1 - 1);
        const unsigned char* row = m0 + (yy - m_lo) * W;
        for (int xx = xa; xx <= xb; ++xx)
          if (row[xx]) { v = true; break; }
      }
    }
    dil[i] = v;
  }
  __syncthreads();
  // ---- erosion, relabel, convert, NaN rule --------------------------------------------------------
  for (int i = tid; i < rows * W; i += kThreads) {
    const int ry = i / W, x = i - ry * W, y = r0 + ry;
    const long gi = base + (long)y * W + x;
    long l = load_label<TEST>(labels_in, label_bytes, gi);
    if (!empty && l > 0) {        // label > 0 implies inside the crop
      bool closed = true;
      for (int dy = -3; dy <= 3 && closed; ++dy) {
        const int yy = y + dy, hw = disk_hw(dy);
        if (yy < y0 || yy >= y1 || x - hw < x0 || x + hw >= x1) { closed = false; break; }
        const unsigned char* row = dil + (yy - d_lo) * W;
        for (int xx = x - hw; xx <= x + hw; ++xx)
          if (!row[xx]) { closed = false; break; }
      }
      if (!closed) l = -30;
    }
    if (TEST) {
      // mask_label_seabed: background pixels below the (padded) seabed -> -50; boundary / fish labels take precedence
      const int cy = (int)tc.centres[2 * b], cx = (int)tc.centres[2 * b + 1];
      const int yd = cy - H / 2 + 1 + y, xd = cx - W / 2 + 1 + x;            // data coordinates of this patch pixel
      if (l == 0 && yd >= 0 && yd < tc.n_range) {
        // zarr reader: the pad shifts the mask down INSIDE the slice the patch asks for (rows from y_top); Echogram:
        // absolute rows >= seabed + pad
        const int y_top = tc.seabed_rule == 0 ? max(cy - H / 2 + 1, 0) : 0;
        if (yd - y_top >= tc.seabed_pad) {
          bool below = false;
          if (tc.seabed) {
            const int xs = xd - tc.seabed_ping0;
            below = xs >= 0 && xs < tc.seabed_pings && (yd - tc.seabed_pad) >= tc.seabed[xs];
          } else if (tc.seabed_mask) {
            // mask_ping0 == CRIMAC_MASK_PER_PATCH: the mask is laid out per patch, [B][W][n_range] (columns the patch
            // has outside the survey are zero rows) -- a batch whose patches lie anywhere in a long survey
            const bool per_patch = tc.mask_ping0 == CRIMAC_MASK_PER_PATCH;
            const int xm = per_patch ? b * W + x : xd - tc.mask_ping0;
            below = xm >= 0 && xm < tc.mask_pings && tc.seabed_mask[(long)xm * tc.n_range + (yd - tc.seabed_pad)];
          }
          if (below) l = -50;
        }
      }
      // mask_label_overlap: the rim shared with the neighbouring patches -> -70, except where the crop left the data
      const int o = tc.overlap;
      if (o > 0 && (y < o || y >= H - o || x < o || x >= W - o) && l != -100) l = -70;
      // remove_nan_inf's label rule (the data transform runs after the label transform)
      if (!isfinite(data[(long)b * C * HW + (long)y * W + x])) l = -100;
    } else if (mode == 1) {
      l = l == 0 ? 0 : l == 27 ? 1 : l == 1 ? 2 : -100;
      bool bad;
      if (aux) bad = aux[gi] & 2;
      else bad = !isfinite(data[(long)b * C * HW + (long)y * W + x]);
      if (bad) l = -100;
    }
    labels_out[gi] = (short)l;
  }
}

// get_extended_label_mask_for_crop (batch/label_transforms/extend_label_masks.py:35-98), the last link of
// define_label_transform_test when eval_mode is 'region' / 'trace' (batch/transforms.py:87-90): a pixel keeps its label
// only inside a (host-extended) school bounding box, everything else becomes `ignore_val`; remove_nan_inf's label rule
// (which the reference applies AFTER the label chain) is re-applied on top.  One workgroup per patch; the boxes are
// filtered against the patch 1024 at a time into an LDS list, every thread carries the keep flags of its 64-pixel column
// share in registers.
constexpr int kBoxRound = 1024;

__global__ __launch_bounds__(kThreads) void extend_mask_kernel(short* __restrict__ labels, const float* __restrict__ data,
                                                               int C, const long long* __restrict__ centres,
                                                               const int* __restrict__ boxes, int n_boxes, int ignore_val,
                                                               int H, int W) {
  __shared__ int list[kBoxRound][4];
  __shared__ int n_list;
  const int b = blockIdx.x, tid = threadIdx.x, HW = H * W;
  // the reference places the crop at centre - shape // 2 (extend_label_masks.py:64), one pixel off the crop's real
  // origin (utils/np.py:378-380) -- restated as is
  const int yul = (int)centres[2 * b] - H / 2, xul = (int)centres[2 * b + 1] - W / 2;
  const int per_thread = (HW + kThreads - 1) / kThreads;
  unsigned long long keep = 0;                             // one bit per pixel of this thread (H * W <= 64 * 1024)
  for (int r0 = 0; r0 < n_boxes; r0 += kBoxRound) {
    if (tid == 0) n_list = 0;
    __syncthreads();
    const int k = r0 + tid;
    if (tid < kBoxRound && k < n_boxes) {
      // rows / columns of the patch the box covers: [max(b0 - yul, 0), min(b1 - yul, H)) x [max(b2 - xul, 0), min(b3 - xul, W))
      const int ya = max(boxes[4 * k] - yul, 0), yb = min(boxes[4 * k + 1] - yul, H);
      const int xa = max(boxes[4 * k + 2] - xul, 0), xb = min(boxes[4 * k + 3] - xul, W);
      if (yb > ya && xb > xa) {
        const int at = atomicAdd(&n_list, 1);
        list[at][0] = ya; list[at][1] = yb; list[at][2] = xa; list[at][3] = xb;
      }
    }
    __syncthreads();
    const int nl = n_list;
    if (nl > 0) {
      for (int j = 0; j < per_thread; ++j) {
        const int i = tid + j * kThreads;
        if (i >= HW) break;
        const int y = i / W, x = i - y * W;
        bool in = false;
        for (int q = 0; q < nl && !in; ++q) in = y >= list[q][0] && y < list[q][1] && x >= list[q][2] && x < list[q][3];
        if (in) keep |= 1ull << j;
      }
    }
    __syncthreads();
  }
  for (int j = 0; j < per_thread; ++j) {
    const int i = tid + j * kThreads;
    if (i >= HW) break;
    const long gi = (long)b * HW + i;
    if (!isfinite(data[(long)b * C * HW + i])) labels[gi] = -100;
    else if (!((keep >> j) & 1)) labels[gi] = (short)ignore_val;
  }
}

}  // namespace

extern "C" int crimac_labels_extend_mask(short* labels, const float* data, int C, const long long* centres,
                                         const int* boxes, int n_boxes, int ignore_val, int B, int H, int W,
                                         void* stream) {
  CRIMAC_REQUIRE(labels && data && centres && (boxes || n_boxes == 0) && n_boxes >= 0 && C > 0,
                 "labels_extend_mask: bad arguments (n_boxes=%d)", n_boxes);
  CRIMAC_REQUIRE(B > 0 && H > 0 && W > 0 && (long)H * W <= 64L * kThreads,
                 "labels_extend_mask: patch of %d x %d (at most 65536 pixels)", H, W);
  CRIMAC_REQUIRE(ignore_val >= -32768 && ignore_val <= 32767, "labels_extend_mask: ignore_val %d", ignore_val);
  hipLaunchKernelGGL(extend_mask_kernel, dim3(B), dim3(kThreads), 0, (hipStream_t)stream, labels, data, C, centres, boxes,
                     n_boxes, ignore_val, H, W);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_refine_labels(const void* labels_in, int label_bytes, const unsigned char* aux_mask,
                                    const float* data, int thr_channel, float thr_lo, float thr_hi, int mode,
                                    short* labels_out, int B, int C, int H, int W, void* stream) {
  CRIMAC_REQUIRE(labels_in && labels_out && (label_bytes == 2 || label_bytes == 4 || label_bytes == 8),
                 "refine_labels: bad label arguments (label_bytes=%d)", label_bytes);
  CRIMAC_REQUIRE((aux_mask != nullptr) != (data != nullptr), "refine_labels: pass exactly one of aux_mask / data");
  CRIMAC_REQUIRE(B > 0 && H > 0 && W > 0 && W <= 1024 && (long)H * W < (1L << 30),
                 "refine_labels: patch of %d x %d does not fit", H, W);
  CRIMAC_REQUIRE(aux_mask || (C > 0 && thr_channel >= 0 && thr_channel < C), "refine_labels: bad channel %d of %d",
                 thr_channel, C);
  CRIMAC_REQUIRE(mode == 0 || mode == 1, "refine_labels: bad mode %d", mode);
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_labels_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const size_t lds = (size_t)(((2 * BAND + 18) * W + 15) & ~15) + 16;
  hipLaunchKernelGGL(refine_labels_kernel<false>, dim3(B, cdiv(H, BAND)), dim3(kThreads), lds, (hipStream_t)stream,
                     labels_in, label_bytes, aux_mask, data, thr_channel, thr_lo, thr_hi, mode, labels_out, C, H, W,
                     TestChain{});
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_labels_test_transform(const void* labels_in, int label_bytes, const float* data, int thr_channel,
                                            float thr_lo, float thr_hi, const long long* centres, const int* seabed,
                                            int seabed_ping0, int seabed_pings, const unsigned char* seabed_mask,
                                            int mask_ping0, int mask_pings, int n_range, int seabed_pad,
                                            int seabed_rule, int overlap, short* labels_out, int B, int C, int H,
                                            int W, void* stream) {
  CRIMAC_REQUIRE(labels_in && labels_out && data && centres && (label_bytes == 2 || label_bytes == 4 || label_bytes == 8),
                 "labels_test_transform: bad arguments (label_bytes=%d)", label_bytes);
  CRIMAC_REQUIRE(B > 0 && H > 0 && W > 0 && W <= 1024 && H % 2 == 0 && W % 2 == 0 && (long)H * W < (1L << 30),
                 "labels_test_transform: patch of %d x %d (even sizes, W <= 1024)", H, W);
  CRIMAC_REQUIRE(C > 0 && thr_channel >= 0 && thr_channel < C, "labels_test_transform: bad channel %d of %d", thr_channel, C);
  CRIMAC_REQUIRE(!(seabed && seabed_mask), "labels_test_transform: give the seabed vector OR the seabed mask");
  CRIMAC_REQUIRE(seabed_rule == 0 || seabed_rule == 1, "labels_test_transform: seabed_rule=%d", seabed_rule);
  CRIMAC_REQUIRE(overlap >= 0 && 2 * overlap < H && 2 * overlap < W && n_range > 0 && seabed_pad >= 0,
                 "labels_test_transform: overlap %d / n_range %d / pad %d", overlap, n_range, seabed_pad);
  static unsigned long long attr_devs = 0;
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_labels_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const size_t lds = (size_t)(((2 * BAND + 18) * W + 15) & ~15) + 16;
  TestChain tc{centres, seabed, seabed_ping0, seabed_pings, seabed_mask, mask_ping0, mask_pings, n_range, seabed_pad,
               seabed_rule, overlap};
  hipLaunchKernelGGL(refine_labels_kernel<true>, dim3(B, cdiv(H, BAND)), dim3(kThreads), lds, (hipStream_t)stream,
                     labels_in, label_bytes, (const unsigned char*)nullptr, data, thr_channel, thr_lo, thr_hi, 2, labels_out,
                     C, H, W, tc);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
