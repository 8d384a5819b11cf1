// Whole-network weight re-layout in ONE launch each way (CDNA4 / gfx950).
//
// Every training step the fp32 master weights change (optim.SGD.step, pipeline.py:178), so the bf16
// operand planes the MFMA kernels read are rebuilt every step, and every weight gradient comes out of
// crimac_wgrad in the packed [tap][f][s] order and has to go back to the torch layout.  Per layer
// these are tiny, badly coalesced transposes (tap-innermost fp32 <-> channel-innermost planes); done
// layer by layer they cost 44 launches and ~0.45 ms per step.  Here one workgroup owns one
// 32 x 32 x taps tile of one layer, moves it through LDS so that both the fp32 side and the packed side
// are accessed in >= 64-byte runs, and finds its layer in a descriptor table passed by value.
#include "common.h"

namespace {

constexpr int kMaxLayers = 32;
constexpr int TILE = 32;

struct LayerDesc {             // mirrors crimac_layer_desc (include/crimac_unet_hip.h)
  const float* w;
  float* grad;
  const float* dw;
  unsigned short* fwd_hi;
  unsigned short* fwd_lo;
  unsigned short* dg_hi;
  unsigned short* dg_lo;
  int kind, Co, Ci, Ci_pad;  // (kind: bit 0 only here; the fragment-major flags travel in `frag`)
  int dw_splits;             // partial slabs of dw to add up (crimac_wgrad_partials); <= 1: dw is the gradient
  long dw_stride;            // floats between slabs
  int frag;                  // bit 0: fwd_hi fragment-major, bit 1: dg_hi fragment-major (common.h wfrag_index)
};

struct Table {
  LayerDesc d[kMaxLayers];
  int first[kMaxLayers + 1];   // first workgroup of each layer; first[n] = total
  int n;
};

// Geometry of a layer seen as [outer][inner][taps] in the torch layout:
//   kind 0 (conv3x3,  w[Co][Ci][3][3]): outer = co, inner = ci, taps = 9
//   kind 1 (upconv2x2, w[Ci][Co][2][2]): outer = ci, inner = co, taps = 4
struct Geo {
  int outer, inner, inner_pad, taps, tiles_inner;
};
__device__ __host__ inline Geo geo_of(int kind, int Co, int Ci, int Ci_pad) {
  Geo g;
  if (kind == 0) { g.outer = Co; g.inner = Ci; g.inner_pad = Ci_pad; g.taps = 9; }
  else { g.outer = Ci; g.inner = Co; g.inner_pad = Co; g.taps = 4; }
  g.tiles_inner = (g.inner_pad + TILE - 1) / TILE;
  return g;
}

__device__ __forceinline__ int find_layer(const Table& tb, int block) {
  int l = 0;
  while (l + 1 < tb.n && block >= tb.first[l + 1]) ++l;
  return l;
}

// two adjacent values -> plane words; plane 0 -> hi[i/2], planes 1.. -> lo[(k-1)*n/2 + i/2]
__device__ __forceinline__ void put_planes2(float v0, float v1, int npl, unsigned short* hi, unsigned short* lo,
                                            long i, long n, int fp16) {
  unsigned short b0 = f2bits16(v0, fp16), b1 = f2bits16(v1, fp16);
  *reinterpret_cast<unsigned int*>(hi + i) = (unsigned int)b0 | ((unsigned int)b1 << 16);
  for (int k = 1; k < npl; ++k) {
    v0 -= bits162f(b0, fp16);
    v1 -= bits162f(b1, fp16);
    b0 = f2bits16(v0, fp16);
    b1 = f2bits16(v1, fp16);
    *reinterpret_cast<unsigned int*>(lo + (long)(k - 1) * n + i) = (unsigned int)b0 | ((unsigned int)b1 << 16);
  }
}

// eight adjacent values -> one 16-byte store per plane (4-byte stores cost the pack kernel 2.5x its traffic time)
__device__ __forceinline__ void put_planes8(float (&v)[8], int npl, unsigned short* hi, unsigned short* lo, long i,
                                            long n, int fp16) {
  unsigned short b[8];
  for (int k = 0; k < npl; ++k) {
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      b[q] = f2bits16(v[q], fp16);
      v[q] -= bits162f(b[q], fp16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = (unsigned int)b[2 * q] | ((unsigned int)b[2 * q + 1] << 16);
    *reinterpret_cast<u32x4*>((k == 0 ? hi : lo + (long)(k - 1) * n) + i) = w;
  }
}

// interleaved plane pairs (CRIMAC_PLANES_INTERLEAVED): 8 adjacent channels col .. col+7 of a row of `rowlen` channels
__device__ __forceinline__ void put_planes8_il(float (&v)[8], unsigned short* buf, long row, int col, int rowlen, int fp16) {
  unsigned short* dst = buf + row * 2L * rowlen + il_pos(col, rowlen);
  put_planes8(v, 2, dst, dst + il_cb(rowlen), 0, 0, fp16);
}

// ... fragment-major (common.h wfrag_index on the row of 2 * rowlen halves): the hi group at the halves-column il_pos(col),
// the lo group one block (32 halves) further
__device__ __forceinline__ void put_planes8_il_frag(float (&v)[8], unsigned short* buf, int t, int r, int R, int col, int rowlen,
                                                    int fp16) {
  const int c = (int)il_pos(col, rowlen);
  unsigned short* hi = buf + wfrag_index(t, r, c, R, 2 * rowlen);
  unsigned short* lo = buf + wfrag_index(t, r, c + il_cb(rowlen), R, 2 * rowlen);
  put_planes8(v, 2, hi, lo, 0, 0, fp16);
}

__global__ __launch_bounds__(256) void pack_layers_kernel(Table tb, int planes_arg) {
  __shared__ float tile[TILE][TILE * 9 + 1];
  const PlaneFmt pf = plane_fmt(planes_arg);
  const int npl = pf.npl;
  const int l = find_layer(tb, blockIdx.x);
  const LayerDesc& d = tb.d[l];
  const Geo g = geo_of(d.kind, d.Co, d.Ci, d.Ci_pad);
  const int tidx = blockIdx.x - tb.first[l];
  const int o0 = (tidx / g.tiles_inner) * TILE, i0 = (tidx % g.tiles_inner) * TILE;
  const int T = g.taps, run = TILE * T;
  const int valid = (g.inner - i0 < TILE ? (g.inner - i0 > 0 ? g.inner - i0 : 0) : TILE) * T;
  // fp32 side: per outer index one contiguous run of (inner x taps)
  // (row by row: a flat index would cost an integer division by the run length per element)
  for (int ol = threadIdx.x >> 6; ol < TILE; ol += 4) {
    const float* src = d.w + ((long)(o0 + ol) * g.inner + i0) * T;
    // (weights and packed gradients are touched once per step: streamed past the caches, common.h load8s)
#if CRIMAC_STREAM_NT
    for (int r = threadIdx.x & 63; r < run; r += 64) tile[ol][r] = r < valid ? __builtin_nontemporal_load(src + r) : 0.f;
#else
    for (int r = threadIdx.x & 63; r < run; r += 64) tile[ol][r] = r < valid ? src[r] : 0.f;
#endif
  }
  __syncthreads();
  const long n = (long)T * g.outer * g.inner_pad;
  // pass A, inner index fastest: conv3x3 forward panel [t][co][ci_pad] / upconv dgrad panel [ab][ci][co]
  unsigned short* a_hi = d.kind == 0 ? d.fwd_hi : d.dg_hi;
  unsigned short* a_lo = d.kind == 0 ? d.fwd_lo : d.dg_lo;
  const int a_fp16 = d.kind == 0 ? pf.fwd_fp16 : pf.dg_fp16, b_fp16 = d.kind == 0 ? pf.dg_fp16 : pf.fwd_fp16;
  const float a_sc = d.kind == 0 ? pf.fwd_scale : pf.dg_scale, b_sc = d.kind == 0 ? pf.dg_scale : pf.fwd_scale;
  if (a_hi && g.inner_pad % 8 == 0) {
    for (int idx = threadIdx.x; idx < T * TILE * (TILE / 8); idx += 256) {
      const int il = (idx % (TILE / 8)) * 8, ol = (idx / (TILE / 8)) % TILE, t = idx / (TILE * TILE / 8);
      if (i0 + il >= g.inner_pad) continue;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tile[ol][(il + q) * T + t] * a_sc;
      if (pf.interleaved && (d.frag & 1)) put_planes8_il_frag(v, a_hi, t, o0 + ol, g.outer, i0 + il, g.inner_pad, a_fp16);
      else if (pf.interleaved) put_planes8_il(v, a_hi, (long)t * g.outer + o0 + ol, i0 + il, g.inner_pad, a_fp16);
      else if (d.frag & 1) put_planes8(v, npl, a_hi, a_lo, wfrag_index(t, o0 + ol, i0 + il, g.outer, g.inner_pad), n, a_fp16);
      else put_planes8(v, npl, a_hi, a_lo, ((long)t * g.outer + o0 + ol) * g.inner_pad + i0 + il, n, a_fp16);
    }
  } else if (a_hi) {
    for (int idx = threadIdx.x; idx < T * TILE * (TILE / 2); idx += 256) {
      const int il = (idx % (TILE / 2)) * 2, ol = (idx / (TILE / 2)) % TILE, t = idx / (TILE * TILE / 2);
      if (i0 + il >= g.inner_pad) continue;
      put_planes2(tile[ol][il * T + t] * a_sc, tile[ol][(il + 1) * T + t] * a_sc, npl, a_hi, a_lo,
                  ((long)t * g.outer + o0 + ol) * g.inner_pad + i0 + il, n, a_fp16);
    }
  }
  // pass B, outer index fastest: conv3x3 dgrad panel [8-t][ci][co] / upconv forward panel [ab][co][ci]
  unsigned short* b_hi = d.kind == 0 ? d.dg_hi : d.fwd_hi;
  unsigned short* b_lo = d.kind == 0 ? d.dg_lo : d.fwd_lo;
  if (b_hi) {              // (outer is a multiple of 32: the 16-byte stores are aligned)
    for (int idx = threadIdx.x; idx < T * TILE * (TILE / 8); idx += 256) {
      const int ol = (idx % (TILE / 8)) * 8, il = (idx / (TILE / 8)) % TILE, t = idx / (TILE * TILE / 8);
      if (i0 + il >= g.inner) continue;
      const int tt = d.kind == 0 ? T - 1 - t : t;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tile[ol + q][il * T + t] * b_sc;
      if (pf.interleaved && (d.frag & 2)) put_planes8_il_frag(v, b_hi, tt, i0 + il, g.inner, o0 + ol, g.outer, b_fp16);
      else if (pf.interleaved) put_planes8_il(v, b_hi, (long)tt * g.inner + i0 + il, o0 + ol, g.outer, b_fp16);
      else if (d.frag & 2) put_planes8(v, npl, b_hi, b_lo, wfrag_index(tt, i0 + il, o0 + ol, g.inner, g.outer), n, b_fp16);
      else put_planes8(v, npl, b_hi, b_lo, ((long)tt * g.inner + i0 + il) * g.outer + o0 + ol, n, b_fp16);
    }
  }
}

// Partial slabs of the weight gradients (crimac_wgrad_partials): layers whose pixel range was split many ways (the
// 64-channel layers at 256 x 256: 512 slabs of 147 KB) are first folded to kFold slabs, slab j += slabs j + kFold,
// j + 2 kFold, ... (in place, fixed order), by a grid that is parallel over elements AND over the kFold groups;
// unpack_layers_kernel then adds up at most kFold slabs per element.  Without this stage the unpack tiles of such a
// layer (4 workgroups) would each walk 512 slabs serially.
constexpr int kFold = 16;
struct FoldTable {
  float* dw[kMaxLayers];
  long stride[kMaxLayers];
  int splits[kMaxLayers];
  int n4[kMaxLayers];          // float4 elements per slab
  int first[kMaxLayers + 1];   // first workgroup (x index) of each layer
  int n;
};
__global__ __launch_bounds__(256) void fold_slabs_kernel(FoldTable tb) {
  int l = 0;
  while (l + 1 < tb.n && (int)blockIdx.x >= tb.first[l + 1]) ++l;
  const int i = (blockIdx.x - tb.first[l]) * 256 + threadIdx.x;
  const int j = blockIdx.y;                                   // group: slabs j, j + kFold, ...
  if (i >= tb.n4[l] || j >= tb.splits[l]) return;
  float* base = tb.dw[l] + 4L * i;
  const long st = tb.stride[l];
  f32x4 acc = *reinterpret_cast<const f32x4*>(base + j * st);
  int k = j + kFold;
  for (; k + 3 * kFold < tb.splits[l]; k += 4 * kFold) {      // four independent loads in flight
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + (long)k * st);
    const f32x4 b = *reinterpret_cast<const f32x4*>(base + (long)(k + kFold) * st);
    const f32x4 c = *reinterpret_cast<const f32x4*>(base + (long)(k + 2 * kFold) * st);
    const f32x4 d = *reinterpret_cast<const f32x4*>(base + (long)(k + 3 * kFold) * st);
    acc += a; acc += b; acc += c; acc += d;
  }
  for (; k < tb.splits[l]; k += kFold) acc += *reinterpret_cast<const f32x4*>(base + (long)k * st);
  *reinterpret_cast<f32x4*>(base + j * st) = acc;
}

__global__ __launch_bounds__(256) void unpack_layers_kernel(Table tb) {
  __shared__ float tile[TILE][TILE * 9 + 1];
  const int l = find_layer(tb, blockIdx.x);
  const LayerDesc& d = tb.d[l];
  const Geo g = geo_of(d.kind, d.Co, d.Ci, d.Ci_pad);
  const int tidx = blockIdx.x - tb.first[l];
  const int o0 = (tidx / g.tiles_inner) * TILE, i0 = (tidx % g.tiles_inner) * TILE;
  const int T = g.taps, run = TILE * T;
  const int nin = g.inner - i0 < TILE ? (g.inner - i0 > 0 ? g.inner - i0 : 0) : TILE;
  // packed side: dw[t][outer][inner_pad] (conv3x3: [t][co][ci_pad], upconv: [ab][ci][co]), inner fastest
#pragma unroll 2
  for (int idx = threadIdx.x; idx < T * TILE * TILE; idx += 256) {
    const int il = idx % TILE, ol = (idx / TILE) % TILE, t = idx / (TILE * TILE);
    if (il < nin) {
      const float* src = d.dw + ((long)t * g.outer + o0 + ol) * g.inner_pad + i0 + il;
      // slabs: four independent partial sums (four loads in flight: one load per loop iteration made this pass a chain of
      // memory latencies), added up in a FIXED order -- reproducible run to run
      const int nsl = d.dw_splits < kFold ? d.dw_splits : kFold;                       // (folded by fold_slabs_kernel)
      float v0 = src[0], v1 = 0.f, v2 = 0.f, v3 = 0.f;
      int sp = 1;
      for (; sp + 3 < nsl; sp += 4) {
        const float a = src[(long)sp * d.dw_stride], b = src[(long)(sp + 1) * d.dw_stride];
        const float c = src[(long)(sp + 2) * d.dw_stride], e = src[(long)(sp + 3) * d.dw_stride];
        v0 += a; v1 += b; v2 += c; v3 += e;
      }
      for (; sp < nsl; ++sp) v0 += src[(long)sp * d.dw_stride];
      tile[ol][il * T + t] = (v0 + v1) + (v2 + v3);
    }
  }
  __syncthreads();
  const int valid = nin * T;
  for (int ol = threadIdx.x >> 6; ol < TILE; ol += 4) {
    float* dst = d.grad + ((long)(o0 + ol) * g.inner + i0) * T;
    for (int r = threadIdx.x & 63; r < valid; r += 64) dst[r] = tile[ol][r];
  }
}

struct HostDesc {              // == crimac_layer_desc
  const float* w;
  float* grad;
  const float* dw;
  void *fwd_hi, *fwd_lo, *dg_hi, *dg_lo;
  int kind, Co, Ci, Ci_pad;
  int dw_splits;
  long dw_stride;
};

// mode 0: pack, 1: unpack
int run_layers(const HostDesc* descs, int n, int mode, int planes_arg, hipStream_t st) {
  const int planes = planes_arg & 15;
  for (int base = 0; base < n; base += kMaxLayers) {
    Table tb;
    tb.n = n - base < kMaxLayers ? n - base : kMaxLayers;
    int total = 0;
    for (int i = 0; i < tb.n; ++i) {
      HostDesc h = descs[base + i];
      const int frag = ((h.kind & CRIMAC_LAYER_FWD_FRAG) ? 1 : 0) | ((h.kind & CRIMAC_LAYER_DG_FRAG) ? 2 : 0);
      CRIMAC_REQUIRE((h.kind & ~(1 | CRIMAC_LAYER_FWD_FRAG | CRIMAC_LAYER_DG_FRAG)) == 0 && h.Co > 0 && h.Ci > 0,
                     "layer %d: bad kind/shape", base + i);
      h.kind &= 1;
      if (frag && mode == 0) {
        const bool ilv = (planes_arg & CRIMAC_PLANES_INTERLEAVED) != 0;
        const int kq = ilv ? 32 : 64;        // (a 64-deep chunk of the row of halves is 32 channels of a plane pair)
        CRIMAC_REQUIRE(h.kind == 0 && (planes == 1 || ilv),
                       "layer %d: fragment-major planes are a single-plane 16-bit or interleaved-pair Conv2d layout", base + i);
        CRIMAC_REQUIRE(!(frag & 1) || (h.Co % 32 == 0 && h.Ci_pad % kq == 0), "layer %d: fragment-major forward plane needs "
                       "Co %% 32 == 0 and Ci_pad %% %d == 0 (Co=%d Ci_pad=%d)", base + i, kq, h.Co, h.Ci_pad);
        CRIMAC_REQUIRE(!(frag & 2) || (h.dg_hi && h.Ci % 32 == 0 && h.Co % kq == 0 && h.Ci_pad == h.Ci),
                       "layer %d: fragment-major input-gradient plane needs Ci %% 32 == 0 and Co %% %d == 0", base + i, kq);
      }
      CRIMAC_REQUIRE(h.Co % TILE == 0 && (h.kind == 0 || h.Ci % TILE == 0),
                     "layer %d: Co (and the transposed convolution's Ci) must be multiples of %d", base + i, TILE);
      CRIMAC_REQUIRE(h.kind == 1 || (h.Ci_pad >= h.Ci && h.Ci_pad % 2 == 0), "layer %d: bad Ci_pad", base + i);
      if (mode == 0) {
        const bool il = (planes_arg & CRIMAC_PLANES_INTERLEAVED) != 0;       // (one buffer holds both planes)
        CRIMAC_REQUIRE(h.w && h.fwd_hi && (planes == 1 || il || h.fwd_lo) && (planes == 1 || il || !h.dg_hi || h.dg_lo),
                       "layer %d: missing weight / plane pointers", base + i);
        CRIMAC_REQUIRE(!il || h.kind == 1 || h.Ci_pad % 8 == 0, "layer %d: interleaved planes need Ci_pad %% 8 == 0", base + i);
        CRIMAC_REQUIRE(h.kind == 1 || !h.dg_hi || h.Ci_pad == h.Ci, "layer %d: dgrad planes need Ci_pad == Ci",
                       base + i);
      } else {
        CRIMAC_REQUIRE(h.dw && h.grad, "layer %d: missing dw / grad pointers", base + i);
      }
      LayerDesc& d = tb.d[i];
      d.w = h.w; d.grad = h.grad; d.dw = h.dw;
      d.fwd_hi = (unsigned short*)h.fwd_hi; d.fwd_lo = (unsigned short*)h.fwd_lo;
      d.dg_hi = (unsigned short*)h.dg_hi; d.dg_lo = (unsigned short*)h.dg_lo;
      d.kind = h.kind; d.Co = h.Co; d.Ci = h.Ci; d.Ci_pad = h.kind == 0 ? h.Ci_pad : h.Ci;
      d.dw_splits = h.dw_splits; d.dw_stride = h.dw_stride;
      d.frag = mode == 0 ? frag : 0;
      if (mode == 1)
        CRIMAC_REQUIRE(h.dw_splits <= 1 || h.dw_stride >= (long)(h.kind == 0 ? 9L * h.Co * h.Ci_pad : 4L * h.Co * h.Ci),
                       "layer %d: dw_stride smaller than one packed gradient", base + i);
      const Geo g = geo_of(d.kind, d.Co, d.Ci, d.Ci_pad);
      tb.first[i] = total;
      total += (g.outer / TILE) * g.tiles_inner;
    }
    tb.first[tb.n] = total;
    if (total == 0) continue;
    if (mode == 1) {
      FoldTable ft;
      ft.n = 0;
      int fb = 0;
      for (int i = 0; i < tb.n; ++i) {
        const LayerDesc& d = tb.d[i];
        if (d.dw_splits <= kFold) continue;
        const long nfl = (long)(d.kind == 0 ? 9 : 4) * d.Co * d.Ci_pad;
        CRIMAC_REQUIRE(nfl % 4 == 0 && d.dw_stride % 4 == 0 && ((uintptr_t)d.dw % 16 == 0),
                       "layer %d: partial slabs must be 16-byte aligned", base + i);
        ft.dw[ft.n] = const_cast<float*>(d.dw); ft.stride[ft.n] = d.dw_stride; ft.splits[ft.n] = d.dw_splits;
        ft.n4[ft.n] = (int)(nfl / 4); ft.first[ft.n] = fb;
        fb += (int)((nfl / 4 + 255) / 256);
        ++ft.n;
      }
      if (ft.n) {
        ft.first[ft.n] = fb;
        hipLaunchKernelGGL(fold_slabs_kernel, dim3(fb, kFold), dim3(256), 0, st, ft);
        CRIMAC_LAUNCH_CHECK();
      }
    }
    if (mode == 0) hipLaunchKernelGGL(pack_layers_kernel, dim3(total), dim3(256), 0, st, tb, planes_arg);
    else hipLaunchKernelGGL(unpack_layers_kernel, dim3(total), dim3(256), 0, st, tb);
    CRIMAC_LAUNCH_CHECK();
  }
  return CRIMAC_OK;
}

}  // namespace

extern "C" int crimac_pack_layers(const crimac_layer_desc* descs, int n_layers, int planes, void* stream) {
  CRIMAC_REQUIRE(descs && n_layers > 0 && planes_arg_ok(planes), "pack_layers: bad arguments (planes=%d)", planes);
  return run_layers(reinterpret_cast<const HostDesc*>(descs), n_layers, 0, planes, (hipStream_t)stream);
}

extern "C" int crimac_unpack_wgrad_layers(const crimac_layer_desc* descs, int n_layers, void* stream) {
  CRIMAC_REQUIRE(descs && n_layers > 0, "unpack_wgrad_layers: bad arguments");
  return run_layers(reinterpret_cast<const HostDesc*>(descs), n_layers, 1, 1, (hipStream_t)stream);
}
