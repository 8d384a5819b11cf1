// Shared device/host helpers for the CDNA4 (gfx950) U-Net kernels.
// All kernels assume wave64 and are written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/crimac_unet_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(T) __attribute__((address_space(3))) T*

// ---- error plumbing (host) -------------------------------------------------------------------
void crimac_set_error(const char* fmt, ...);
#define CRIMAC_REQUIRE(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      crimac_set_error(__VA_ARGS__);              \
      return CRIMAC_ERR_INVALID;                  \
    }                                             \
  } while (0)
#define CRIMAC_LAUNCH_CHECK()                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      crimac_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                       hipGetErrorString(e_));                             \
      return CRIMAC_ERR_LAUNCH;                                            \
    }                                                                      \
  } while (0)

// ---- bf16 helpers (device) -------------------------------------------------------------------
// Round-to-nearest-even via the hardware cast (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ float bfbits2f(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f2bfbits(float x) {
  bf16_t b = (bf16_t)x;
  return *reinterpret_cast<unsigned short*>(&b);
}
// hi/lo split of an fp32 value into two bf16: x ~= hi + lo with |x - hi - lo| <= 2^-17 |x|.
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  hi = f2bfbits(x);
  lo = f2bfbits(x - bfbits2f(hi));
}

// ---- 16-bit element traits: the two storage / MFMA operand types of the 16-bit modes -------------------------
// Fragments travel as raw 128-bit registers (typed bf16x8 for historical reasons: LDS reads, DMAs and inline-asm
// loads do not care what the 16 bits mean); only the conversion to / from fp32 and the MFMA opcode depend on T16.
template <typename T16> struct E16;
template <> struct E16<bf16_t> {
  __device__ static __forceinline__ unsigned short bits(float x) { return f2bfbits(x); }
  __device__ static __forceinline__ float val(unsigned short b) { return bfbits2f(b); }
  __device__ static __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct E16<half_t> {
  __device__ static __forceinline__ unsigned short bits(float x) {
    const half_t h = (half_t)x;                   // round-to-nearest-even (v_cvt_f16_f32); overflow -> inf
    return *reinterpret_cast<const unsigned short*>(&h);
  }
  __device__ static __forceinline__ float val(unsigned short b) { return (float)*reinterpret_cast<const half_t*>(&b); }
  __device__ static __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  }
};

// Runtime-selected element type (weight packing): `fp16` != 0 -> IEEE half, else bf16.  The `planes` argument of the
// pack entry points carries it in bit 4 (CRIMAC_PLANES_FP16).
__device__ __forceinline__ unsigned short f2bits16(float x, int fp16) { return fp16 ? E16<half_t>::bits(x) : f2bfbits(x); }
__device__ __forceinline__ float bits162f(unsigned short b, int fp16) { return fp16 ? E16<half_t>::val(b) : bfbits2f(b); }

// Split 8 fp32 values (two 16-byte registers) into NPL bf16 planes of 8 values each:
// x = p0 + p1 (+ p2) up to 2^-17 (2^-25) relative.  Each subtraction is exact in fp32.
template <int NPL, typename P16 = bf16_t>
__device__ __forceinline__ void split8(const u32x4& r0, const u32x4& r1, u32x4 (&pl)[NPL]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float f[2] = {__uint_as_float(j < 2 ? r0[2 * j] : r1[2 * j - 4]),
                  __uint_as_float(j < 2 ? r0[2 * j + 1] : r1[2 * j - 3])};
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const unsigned short b0 = E16<P16>::bits(f[0]), b1 = E16<P16>::bits(f[1]);
      pl[k][j] = (unsigned)b0 | ((unsigned)b1 << 16);
      f[0] -= E16<P16>::val(b0);
      f[1] -= E16<P16>::val(b1);
    }
  }
}

// D += A * B with A, B given as NPL bf16 planes: all plane products of total order < NPL, smallest
// terms first (NPL 1: 1 MFMA, 2: 3 MFMAs, 3: 6 MFMAs).
template <int NPL, typename P16 = bf16_t>
__device__ __forceinline__ void mfma_planes(const bf16x8 (&a)[NPL], const bf16x8 (&b)[NPL], f32x16& acc) {
#pragma unroll
  for (int s = NPL - 1; s >= 0; --s)
#pragma unroll
    for (int i = 0; i <= s; ++i)
      acc = E16<P16>::mfma32(a[i], b[s - i], acc);
}

// MFMA operand (plane) element type of an activation storage type: 16-bit storage is its own plane; fp32 storage
// is split into bf16 planes.
template <typename TA> struct PlaneOf { using type = TA; };
template <> struct PlaneOf<float> { using type = bf16_t; };
struct hp_t;
template <> struct PlaneOf<hp_t> { using type = half_t; };      // plane pairs (below): fp16 hi / lo

// Number of operand planes of a precision mode.
__host__ __device__ constexpr int planes_of(int prec) {
  return (prec == 0 || prec == 3) ? 1 : ((prec == 1 || prec == 4 || prec == 5) ? 2 : 3);
}

// ---- fp16 plane pairs ("hp" storage, CRIMAC_PREC_H3P) -----------------------------------------------------------
// A tensor of this storage type has the ADDRESSING of an fp32 tensor -- 4 bytes per element, pixel stride `ld`
// elements, channel slices at 4 * c bytes -- but every aligned group of 8 channels holds
//     [8 x fp16 hi][8 x fp16 lo]      value = hi + lo   (hi = rn16(x), lo = rn16(x - hi): 22 significant bits)
// i.e. the operand planes of the 3-MFMA product (hi*hi + hi*lo + lo*hi) are split ONCE, by the kernel that produces
// the tensor, and every consumer takes them as they are: an MFMA kernel sees a 16-bit tensor with twice the channels
// (K' = 2K; a 32-channel chunk = 64 halves = 128 bytes per pixel, moved global -> LDS by LDS-DMA like a bf16 chunk)
// and an elementwise kernel sees 32-byte groups it loads / stores with load8 / store8 below.  Because the addressing
// is that of fp32, fp32 and plane-pair tensors may share one buffer (the two halves of a concat buffer).
struct hp_t { unsigned int raw; };          // never used as a value: only through load8 / store8 / storage_round
static_assert(sizeof(hp_t) == 4, "hp_t must address like float");

// value -> what a tensor of storage type T holds for it
template <typename T> __device__ __forceinline__ float storage_round(float x) { return (float)(T)x; }
template <> __device__ __forceinline__ float storage_round<float>(float x) { return x; }
template <> __device__ __forceinline__ float storage_round<hp_t>(float x) {
  const half_t h = (half_t)x;
  const half_t l = (half_t)(x - (float)h);
  return (float)h + (float)l;               // exact: hi + lo spans at most 23 bits
}
// `planes` argument of the packing entry points (crimac_unet_hip.h, CRIMAC_PLANES_*)
struct PlaneFmt {
  int npl, fwd_fp16, dg_fp16;
  float fwd_scale, dg_scale;
  int interleaved;            // CRIMAC_PLANES_INTERLEAVED: plane k of channel c of a row of K channels at il_pos(c, K) + k * CB
};
__host__ __device__ inline PlaneFmt plane_fmt(int planes_arg) {
  PlaneFmt f;
  f.npl = planes_arg & 15;
  f.fwd_fp16 = (planes_arg >> 4) & 1;
  f.dg_fp16 = (planes_arg >> 5) & 1;
  f.interleaved = (planes_arg >> 6) & 1;
  f.fwd_scale = (float)(1 << ((planes_arg >> 8) & 255));
  f.dg_scale = ((planes_arg >> 7) & 1) ? f.fwd_scale : 1.f;
  return f;
}
__host__ inline bool planes_arg_ok(int planes_arg) {
  const int npl = planes_arg & 15, sh = (planes_arg >> 8) & 255;
  if ((planes_arg & 0x40) && npl != 2) return false;           // interleaved rows are plane PAIRS
  return npl >= 1 && npl <= 3 && (planes_arg >> 17) == 0 && sh <= 16;      // (bit 16: CRIMAC_PLANES_FWD_FRAG)
}
// Fragment-major weight plane (CRIMAC_EPI_WFRAG, CRIMAC_LAYER_*_FRAG): position (in halves) of the 8 columns c .. c + 7
// (c % 8 == 0) of row r at tap t in a plane of R rows x K columns.  A (tap, 32-row block, 64-column chunk) is 4 KB: four
// fragments [k-step ks2][row half nb] of 1 KB each, lane (c % 32) / 8 * 16 + r % 16 of a fragment at 16 bytes per lane -- what
// conv3x3_wch_kernel's lane (fr = lane & 15, fq = lane >> 4) loads into f.b[ks2 * 2 + nb].
__host__ __device__ inline long wfrag_index(int t, int r, int c, int R, int K) {
  const long blk = ((long)t * (R >> 5) + (r >> 5)) * (K >> 6) + (c >> 6);
  return ((blk * 4 + (((c & 63) >> 5) << 1) + ((r & 31) >> 4)) * 64 + (((c & 31) >> 3) << 4) + (r & 15)) * 8 + (c & 7);
}

// Interleaved plane-pair rows (CRIMAC_PLANES_INTERLEAVED): a row of K channels is 2K halves, block size CB = min(K, 32);
// channel c of plane k sits at (c / CB) * 2CB + k * CB + c % CB.
__host__ __device__ constexpr int il_cb(int K) { return K < 32 ? K : 32; }
__host__ __device__ inline long il_pos(int c, int K) {
  const int cb = il_cb(K);
  return (long)(c / cb) * 2 * cb + c % cb;
}

// Storage-type dispatch of an entry point: `T` is bf16_t (CRIMAC_PREC_BF16), half_t (CRIMAC_PREC_FP16) or float
// (the fp32-storage modes) inside the statement.  CRIMAC_PREC_H3P: `T` is the type of the tensors that stay fp32 in
// that mode (convolution outputs, activation gradients) -- float; entry points with plane-pair operands dispatch on
// the precision themselves.
#define CRIMAC_FOR_STORAGE(prec, T, ...)                                  \
  do {                                                                    \
    if ((prec) == CRIMAC_PREC_BF16) { using T = bf16_t; __VA_ARGS__; }    \
    else if ((prec) == CRIMAC_PREC_FP16) { using T = half_t; __VA_ARGS__; } \
    else { using T = float; __VA_ARGS__; }                                \
  } while (0)
// Two storage types: TF for the tensors that are fp32 in H3P (conv outputs y, activation gradients da), TP for the
// MFMA operands of that mode (activations a, output gradients dy): hp_t.  All other modes: both the mode's type.
#define CRIMAC_FOR_STORAGE2(prec, TF, TP, ...)                                            \
  do {                                                                                    \
    if ((prec) == CRIMAC_PREC_BF16) { using TF = bf16_t; using TP = bf16_t; __VA_ARGS__; } \
    else if ((prec) == CRIMAC_PREC_FP16) { using TF = half_t; using TP = half_t; __VA_ARGS__; } \
    else if ((prec) == CRIMAC_PREC_H3P) { using TF = float; using TP = hp_t; __VA_ARGS__; } \
    else if ((prec) == CRIMAC_PREC_H3F_BWD) { using TF = float; using TP = half_t; __VA_ARGS__; } \
    else { using TF = float; using TP = float; __VA_ARGS__; }                             \
  } while (0)

// Activation element traits: T = bf16_t (16-bit storage) or float (fp32 storage).
template <typename T> struct ActT;
template <> struct ActT<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;  // elements per 16-byte access
};
template <> struct ActT<half_t> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;
};
template <> struct ActT<float> {
  static constexpr int kBytes = 4;
  static constexpr int kVec = 4;
};
template <> struct ActT<hp_t> {
  static constexpr int kBytes = 4;
  static constexpr int kVec = 4;
};

// Load/store 8 consecutive channels as fp32 regardless of storage type.
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  u16x8 r = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = bfbits2f(r[i]);
}
__device__ __forceinline__ void load8(const half_t* p, float (&v)[8]) {
  half8 r = *reinterpret_cast<const half8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
// plane pairs: 8 channels = [8 hi][8 lo] (32 bytes, 16-byte aligned like the two halves of an fp32 group)
__device__ __forceinline__ void hp_join(const u32x4& h, const u32x4& l, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = E16<half_t>::val((unsigned short)(h[j] & 0xffffu)) + E16<half_t>::val((unsigned short)(l[j] & 0xffffu));
    v[2 * j + 1] = E16<half_t>::val((unsigned short)(h[j] >> 16)) + E16<half_t>::val((unsigned short)(l[j] >> 16));
  }
}
__device__ __forceinline__ void hp_split(const float (&v)[8], u32x4& h, u32x4& l) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned short h0 = E16<half_t>::bits(v[2 * j]), h1 = E16<half_t>::bits(v[2 * j + 1]);
    const unsigned short l0 = E16<half_t>::bits(v[2 * j] - E16<half_t>::val(h0));
    const unsigned short l1 = E16<half_t>::bits(v[2 * j + 1] - E16<half_t>::val(h1));
    h[j] = (unsigned)h0 | ((unsigned)h1 << 16);
    l[j] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
}
__device__ __forceinline__ void load8(const hp_t* p, float (&v)[8]) {
  const u32x4 h = *reinterpret_cast<const u32x4*>(p);
  const u32x4 l = *(reinterpret_cast<const u32x4*>(p) + 1);
  hp_join(h, l, v);
}
// The same load for data a streaming kernel reads exactly once (non-temporal: it does not displace what the MFMA
// kernels running beside it keep in L2 / the Infinity Cache).  bn_bwd_apply with it: 161 -> 144 us at level 0, and
// -0.8 % on the whole step in the same call (-DCRIMAC_STREAM_NT=0 builds the plain form for A/B runs).
#ifndef CRIMAC_STREAM_NT
#define CRIMAC_STREAM_NT 1
#endif
template <typename T>
__device__ __forceinline__ void load8s(const T* p, float (&v)[8]) {
#if CRIMAC_STREAM_NT
  if constexpr (__is_same(T, hp_t)) {
    const u32x4 h = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    const u32x4 l = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    hp_join(h, l, v);
  } else if constexpr (sizeof(T) == 2) {
    const u16x8 r = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
    load8(reinterpret_cast<const T*>(&r), v);
  } else {
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
#else
  load8(p, v);
#endif
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  u16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = f2bfbits(v[i]);
  *reinterpret_cast<u16x8*>(p) = r;
}
__device__ __forceinline__ void store8(half_t* p, const float (&v)[8]) {
  half8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (half_t)v[i];
  *reinterpret_cast<half8*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

__device__ __forceinline__ void store8(hp_t* p, const float (&v)[8]) {
  u32x4 h, l;
  hp_split(v, h, l);
  *reinterpret_cast<u32x4*>(p) = h;
  *(reinterpret_cast<u32x4*>(p) + 1) = l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One-time per-DEVICE setup (hipFuncSetAttribute applies to the current device only): true the first time the
// calling site runs with device d current.  `mask` is a static of the call site, bit d = done on device d.
static inline bool crimac_first_use_on_device(unsigned long long* mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const unsigned long long bit = 1ull << dev;
  return (__atomic_fetch_or(mask, bit, __ATOMIC_ACQ_REL) & bit) == 0;
}
// Compute units of the current device (cached per device).
static inline int crimac_cu_count() {
  static int ncu[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&ncu[dev], __ATOMIC_RELAXED);
  if (n <= 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    __atomic_store_n(&ncu[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// ---- in-kernel clock diagnostic (MI355X_MICROARCH.md, DVFS give-back (6)) ---------------------------------
// -DCRIMAC_DIAG_CLOCK builds (tools/diag_clock.py, never the shipped library): a kernel stamps s_memtime (shader
// cycles) and s_memrealtime (100 MHz) around its main loop and thread 0 of each workgroup stores the two
// differences into a buffer of the translation unit that no other code reads; clock = d_memtime / d_memrealtime
// x 100 MHz.  In the normal build the macros expand to nothing.
#ifdef CRIMAC_DIAG_CLOCK
#define CRIMAC_DIAG_SLOTS 4096
#define CRIMAC_DIAG_DECLARE(name)                                                                   \
  __device__ unsigned long long name##_buf[2 * CRIMAC_DIAG_SLOTS];                                  \
  extern "C" int name##_read(unsigned long long* host_out) {                                        \
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(name##_buf), sizeof(unsigned long long) * 2 * CRIMAC_DIAG_SLOTS); \
  }
#define CRIMAC_DIAG_STAMP(t, r)                                                                     \
  unsigned long long t, r;                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory"); \
  __builtin_amdgcn_sched_barrier(0);
#define CRIMAC_DIAG_STORE(name, t0, r0, t1, r1)                                                     \
  if (threadIdx.x == 0) {                                                                           \
    name##_buf[2 * (blockIdx.x % CRIMAC_DIAG_SLOTS)] = (t1) - (t0);                                 \
    name##_buf[2 * (blockIdx.x % CRIMAC_DIAG_SLOTS) + 1] = (r1) - (r0);                             \
  }
#else
#define CRIMAC_DIAG_DECLARE(name)
#define CRIMAC_DIAG_STAMP(t, r)
#define CRIMAC_DIAG_STORE(name, t0, r0, t1, r1)
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
