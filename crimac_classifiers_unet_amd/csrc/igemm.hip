// Implicit-GEMM convolution on MFMA for CDNA4 (gfx950), NHWC activations.
//
// One kernel template covers every dense contraction of the U-Net forward and input-gradient path
// (reference ops: nn.Conv2d 3x3 pad 1, unet.py:35-44; nn.ConvTranspose2d k2 s2, unet.py:47-49):
//
//   GEMM rows    m = output pixel (b, oy, ox) on the row grid [B][Ho][Wo]
//   GEMM cols    n = output channel (or (a,b,o) for the transposed conv)
//   contraction  k = (tap t, input channel c); tap t reads input pixel
//                    (oy*stride + t/tw - pad, ox*stride + t%tw - pad), zero outside the image
//
//   conv3x3 fwd      : ntaps 9, tw 3, pad 1, stride 1, W[t][co][ci]
//   conv3x3 dgrad    : same with flipped/transposed weights W'[t][ci][co] = W[co][ci][8-t]
//   upconv2x2 fwd    : ntaps 1, N = 4*Cout ordered (a,b,o), epilogue scatters to (2oy+a, 2ox+b)
//   upconv2x2 dgrad  : ntaps 4, tw 2, pad 0, stride 2 (gathers the 2x2 fine pixels), N = Cin
//
// Tile: 128 (pixels) x BN (channels) per 256-thread workgroup, 4 waves as 2x2, each wave
// 64 x BN/2 via v_mfma_f32_32x32x16_bf16.  A and B tiles are staged through registers into
// XOR-swizzled LDS images (conflict-free ds_read_b128 fragment reads), with the next step's global
// loads in flight during the MFMAs.
//
// Precision: TA = bf16_t  -> bf16 operands, one MFMA per product (throughput mode);
//            TA = float   -> fp32 activations split on the fly into bf16 hi+lo, weights pre-split;
//                            3 MFMAs (hi*hi + hi*lo + lo*hi), fp32 accumulate: ~2^-16 relative
//                            per product, the parity mode (bf16 alone misses 1e-3).
#include <stdlib.h>

#include "common.h"

namespace {

struct IgemmParams {
  const void* in;
  long in_ld;
  int B, Hi, Wi, Ho, Wo;
  int Cin, N;
  int ntaps, tw, pad, stride;
  const unsigned short* w_hi;
  const unsigned short* w_lo;
  const float* bias;
  int bias_mod;
  void* out;
  long out_ld;
  int relu;
  int cout_up;  // UPSCATTER: columns per (a,b) group
  long M;
  float acc_scale;   // F32H3: 2^-WSHIFT (forward planes hold w * 2^WSHIFT); else 1
};

constexpr int BM = 128;

template <int BK> struct Swz {
  static constexpr int RB = BK * 2;          // row bytes
  static constexpr int UPR = BK / 8;         // 16-byte units per row
  static constexpr int RPL = 256 / RB;       // rows per 256-byte bank line
  __device__ static __forceinline__ int off(int row, int u) {
    return row * RB + ((u ^ ((row / RPL) % UPR)) << 4);
  }
};

template <typename TA, int NPL, int BN, int BK, int OUT_MODE, typename P16 = typename PlaneOf<TA>::type>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  constexpr bool X3 = sizeof(TA) == 4;     // fp32 activations, split into NPL bf16 planes
  static_assert(X3 ? (NPL == 2 || NPL == 3) : NPL == 1, "bf16 -> 1 plane, fp32 -> 2 or 3 planes");
  constexpr int UPR = BK / 8;
  constexpr int NU_A = (BM * UPR + 255) / 256;
  constexpr int NU_B = (BN * UPR + 255) / 256;
  constexpr bool B_GUARD = (BN * UPR) % 256 != 0;   // BN*UPR < 256: only some threads stage B
  constexpr int NT = BN / 64;       // 32-col MFMA tiles per wave
  constexpr int KS = BK / 16;       // MFMA k-steps per LDS tile
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  static_assert((BM * UPR) % 256 == 0, "A tile must be a whole number of 256-thread passes");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto sA = [&](int pl) { return smem + pl * A_BYTES; };
  auto sB = [&](int pl) { return smem + NPL * A_BYTES + pl * B_BYTES; };
  const long w_plane = (long)p.ntaps * p.N * p.Cin;      // elements per weight plane

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware tile order: workgroups that share an XCD (same id % 8) get consecutive tiles so the
  // A panel of an M-tile stays in that XCD's L2 while its N-tiles are swept.
  const int tilesN = p.N / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int tile_m = bid / tilesN, tile_n = bid % tilesN;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread staging coordinates --------------------------------------------------------
  int a_oy[NU_A], a_ox[NU_A], a_row[NU_A], a_u[NU_A];
  long a_img[NU_A];  // pixel index of (b, 0, 0) on the input grid, or -1 if the row is beyond M
#pragma unroll
  for (int i = 0; i < NU_A; ++i) {
    const int q = tid + 256 * i;
    a_row[i] = q / UPR;
    a_u[i] = q % UPR;
    const long m = m0 + a_row[i];
    if (m < p.M) {
      const int ox = (int)(m % p.Wo);
      const long t = m / p.Wo;
      a_oy[i] = (int)(t % p.Ho);
      a_ox[i] = ox;
      a_img[i] = (t / p.Ho) * (long)p.Hi * p.Wi;
    } else {
      a_oy[i] = 0; a_ox[i] = 0; a_img[i] = -1;
    }
  }

  u32x4 ra[NU_A][X3 ? 2 : 1];
  u32x4 rb[NU_B][NPL];
  const int kchunks = p.Cin / BK;
  const int nsteps = kchunks * p.ntaps;

  auto load_step = [&](int s) {
    const int kc = s / p.ntaps, t = s % p.ntaps;
    const int ty = t / p.tw - p.pad, tx = t % p.tw - p.pad;
#pragma unroll
    for (int i = 0; i < NU_A; ++i) {
      const int iy = a_oy[i] * p.stride + ty, ix = a_ox[i] * p.stride + tx;
      const bool ok = a_img[i] >= 0 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      if (ok) {
        const TA* src = reinterpret_cast<const TA*>(p.in) +
                        (a_img[i] + (long)iy * p.Wi + ix) * p.in_ld + kc * BK + a_u[i] * 8;
        ra[i][0] = *reinterpret_cast<const u32x4*>(src);
        if constexpr (X3) ra[i][1] = *reinterpret_cast<const u32x4*>(src + 4);
      } else {
        ra[i][0] = u32x4{0, 0, 0, 0};
        if constexpr (X3) ra[i][1] = u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int i = 0; i < NU_B; ++i) {
      const int q = tid + 256 * i;
      if (B_GUARD && q >= BN * UPR) continue;
      const int row = q / UPR, u = q % UPR;
      const long off = ((long)t * p.N + n0 + row) * p.Cin + kc * BK + u * 8;
      rb[i][0] = *reinterpret_cast<const u32x4*>(p.w_hi + off);
#pragma unroll
      for (int k = 1; k < NPL; ++k)
        rb[i][k] = *reinterpret_cast<const u32x4*>(p.w_lo + (long)(k - 1) * w_plane + off);
    }
  };

  auto store_step = [&]() {
#pragma unroll
    for (int i = 0; i < NU_A; ++i) {
      const int o = Swz<BK>::off(a_row[i], a_u[i]);
      if constexpr (X3) {
        u32x4 pl[NPL];
        split8<NPL, P16>(ra[i][0], ra[i][1], pl);
#pragma unroll
        for (int k = 0; k < NPL; ++k) *reinterpret_cast<u32x4*>(sA(k) + o) = pl[k];
      } else {
        *reinterpret_cast<u32x4*>(sA(0) + o) = ra[i][0];
      }
    }
#pragma unroll
    for (int i = 0; i < NU_B; ++i) {
      const int q = tid + 256 * i;
      if (B_GUARD && q >= BN * UPR) continue;
      const int o = Swz<BK>::off(q / UPR, q % UPR);
#pragma unroll
      for (int k = 0; k < NPL; ++k) *reinterpret_cast<u32x4*>(sB(k) + o) = rb[i][k];
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;

  load_step(0);
  for (int s = 0; s < nsteps; ++s) {
    store_step();
    __syncthreads();
    if (s + 1 < nsteps) load_step(s + 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[2][NPL], bfr[NT][NPL];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = Swz<BK>::off(wr * 64 + i * 32 + fr, 2 * ks + fh);
#pragma unroll
        for (int k = 0; k < NPL; ++k) af[i][k] = *reinterpret_cast<const bf16x8*>(sA(k) + o);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int o = Swz<BK>::off(wc * (BN / 2) + j * 32 + fr, 2 * ks + fh);
#pragma unroll
        for (int k = 0; k < NPL; ++k) bfr[j][k] = *reinterpret_cast<const bf16x8*>(sB(k) + o);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mfma_planes<NPL, P16>(af[i], bfr[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: bias (+ReLU) in registers, tile staged through LDS, coalesced 16-byte stores -----
  constexpr int STAGE_PITCH = BN * (int)sizeof(TA) + 16;
  unsigned char* stage = smem;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = wc * (BN / 2) + j * 32 + fr;
    const float bv = p.bias ? p.bias[(n0 + col) % p.bias_mod] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        float v = acc[i][j][r] * p.acc_scale + bv;
        if (p.relu) v = fmaxf(v, 0.f);
        *reinterpret_cast<TA*>(stage + row * STAGE_PITCH + col * (int)sizeof(TA)) = (TA)v;
      }
  }
  __syncthreads();
  {
    constexpr int CPR = BN / 8;
    constexpr int RPP = 256 / CPR;
    const int c8 = tid % CPR, r0 = tid / CPR;
    const int n = n0 + c8 * 8;
    int up_o = 0, up_a = 0, up_b = 0;
    if constexpr (OUT_MODE == 1) {
      const int ab = n / p.cout_up;          // 8 consecutive columns never straddle an (a,b) group
      up_o = n % p.cout_up;
      up_a = ab >> 1;
      up_b = ab & 1;
    }
    TA* outp = reinterpret_cast<TA*>(p.out);
#pragma unroll
    for (int rr = 0; rr < BM / RPP; ++rr) {
      const int row = r0 + rr * RPP;
      const long m = m0 + row;
      if (m >= p.M) continue;
      long o;
      if constexpr (OUT_MODE == 1) {
        const int ox = (int)(m % p.Wo);
        const long t = m / p.Wo;
        const int oy = (int)(t % p.Ho);
        const long bb = t / p.Ho;
        o = ((bb * (2L * p.Ho) + 2 * oy + up_a) * (2L * p.Wo) + 2 * ox + up_b) * p.out_ld + up_o;
      } else {
        o = m * p.out_ld + n;
      }
      const TA* sp = reinterpret_cast<const TA*>(stage + row * STAGE_PITCH) + c8 * 8;
      if constexpr (X3) {
        *reinterpret_cast<f32x4*>(outp + o) = *reinterpret_cast<const f32x4*>(sp);
        *reinterpret_cast<f32x4*>(outp + o + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
      } else {
        *reinterpret_cast<u32x4*>(outp + o) = *reinterpret_cast<const u32x4*>(sp);
      }
    }
  }
}

template <typename TA, int NPL, int BN, int BK, int OUT_MODE, typename P16 = typename PlaneOf<TA>::type>
int launch(const IgemmParams& p, hipStream_t st) {
  const int tilesM = cdiv(p.M, BM), tilesN = p.N / BN;
  size_t lds = (size_t)(BM + BN) * BK * 2 * NPL;
  const size_t stage = (size_t)BM * (BN * sizeof(TA) + 16);
  if (stage > lds) lds = stage;      // the epilogue staging tile reuses the operand buffers
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<TA, NPL, BN, BK, OUT_MODE, P16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((igemm_kernel<TA, NPL, BN, BK, OUT_MODE, P16>), dim3(tilesM * tilesN), dim3(256), lds, st, p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

template <typename TA, int NPL, int OUT_MODE, typename P16 = typename PlaneOf<TA>::type>
int dispatch(const IgemmParams& p, hipStream_t st) {
  const bool n128 = (p.N % 128) == 0;
  if (p.Cin % 64 == 0 && NPL < 3) {
    return n128 ? launch<TA, NPL, 128, 64, OUT_MODE, P16>(p, st) : launch<TA, NPL, 64, 64, OUT_MODE, P16>(p, st);
  } else if (p.Cin % 32 == 0) {
    return n128 ? launch<TA, NPL, 128, 32, OUT_MODE, P16>(p, st) : launch<TA, NPL, 64, 32, OUT_MODE, P16>(p, st);
  } else {
    return n128 ? launch<TA, NPL, 128, 16, OUT_MODE, P16>(p, st) : launch<TA, NPL, 64, 16, OUT_MODE, P16>(p, st);
  }
}

}  // namespace

// upconv.hip: channel-split LDS-DMA kernel for the two bf16 transposed-convolution contractions
bool crimac_upconv_wch_ok(int ntaps, long in_bytes, int K, int N, int cout_up, long out_ld);
int crimac_upconv_wch_16(int ntaps, const void* in, long in_ld, int B, int H, int W, int K, int N, const void* w,
                         const float* bias, int cout_up, void* out, long out_ld, hipStream_t st, int fp16);
bool crimac_upconv_wch_hp_ok(int ntaps, long in_bytes, int K, int N, int cout_up, long out_ld);
int crimac_upconv_wch_hp(int ntaps, const void* in, long in_ld, int B, int H, int W, int K, int N, const void* w,
                         const float* bias, int cout_up, void* out, long out_ld, hipStream_t st, int out_planes);

extern "C" int crimac_igemm_conv(int prec, const void* in, long in_ld, int B, int Hi, int Wi, int Ho,
                                 int Wo, int Cin, int N, int ntaps, int tw, int pad, int stride,
                                 const void* w_hi, const void* w_lo, const float* bias, int bias_mod,
                                 void* out, long out_ld, int relu, int out_mode, int cout_up,
                                 void* stream) {
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, "igemm: bad precision %d", prec);
  CRIMAC_REQUIRE(Cin > 0 && Cin % 16 == 0, "igemm: Cin=%d must be a positive multiple of 16", Cin);
  CRIMAC_REQUIRE(N > 0 && N % 64 == 0, "igemm: N=%d must be a positive multiple of 64", N);
  CRIMAC_REQUIRE(in_ld >= Cin && in_ld % 8 == 0, "igemm: in_ld=%ld must be >= Cin and a multiple of 8", in_ld);
  CRIMAC_REQUIRE(ntaps >= 1 && tw >= 1 && ntaps % tw == 0 && stride >= 1, "igemm: bad tap geometry");
  CRIMAC_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "igemm: bad grid");
  CRIMAC_REQUIRE(in && w_hi && out, "igemm: null pointer");
  const bool is16 = prec == CRIMAC_PREC_BF16 || prec == CRIMAC_PREC_FP16;
  const int out_planes = (relu & CRIMAC_EPI_OUT_PLANES) != 0;
  relu &= CRIMAC_EPI_RELU;
  CRIMAC_REQUIRE(!out_planes || prec == CRIMAC_PREC_H3P, "igemm: plane-pair output is an H3P option");
  CRIMAC_REQUIRE(is16 || w_lo || prec == CRIMAC_PREC_H3P, "igemm: split precisions need the low weight plane(s)");
  CRIMAC_REQUIRE(out_mode == 0 || (out_mode == 1 && cout_up > 0 && N == 4 * cout_up),
                 "igemm: bad output mode / cout_up");
  CRIMAC_REQUIRE(out_mode == 1 ? out_ld >= cout_up : out_ld >= N, "igemm: out_ld too small");
  CRIMAC_REQUIRE(out_ld % 8 == 0 && (out_mode == 0 || cout_up % 8 == 0),
                 "igemm: out_ld (and cout_up) must be multiples of 8 for the 16-byte stores");
  IgemmParams p;
  p.in = in; p.in_ld = in_ld; p.B = B; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo;
  p.Cin = Cin; p.N = N; p.ntaps = ntaps; p.tw = tw; p.pad = pad; p.stride = stride;
  p.w_hi = (const unsigned short*)w_hi; p.w_lo = (const unsigned short*)w_lo;
  p.bias = bias; p.bias_mod = bias_mod > 0 ? bias_mod : N;
  p.out = out; p.out_ld = out_ld; p.relu = relu; p.cout_up = cout_up;
  p.M = (long)B * Ho * Wo;
  p.acc_scale = prec == CRIMAC_PREC_F32H3 ? 1.f / (float)(1 << CRIMAC_F32H3_WSHIFT) : 1.f;
  hipStream_t st = (hipStream_t)stream;
  if (prec == CRIMAC_PREC_H3P) {
    // plane-pair tensors: only the two transposed-convolution contractions exist in this mode (upconv.hip); forward:
    // plane-pair output (it feeds the decoder's first convolution), input gradient: fp32
    const long in_bytes = (((long)B * Hi * Wi - 1) * in_ld + Cin) * 4;
    const bool fwd = out_mode == 1 && ntaps == 1 && stride == 1 && pad == 0 && Hi == Ho && Wi == Wo &&
                     (!bias || bias_mod == cout_up) && out_planes;
    const bool dgr = out_mode == 0 && ntaps == 4 && tw == 2 && stride == 2 && pad == 0 && Hi == 2 * Ho &&
                     Wi == 2 * Wo && !bias && !out_planes;
    CRIMAC_REQUIRE(!relu && (fwd || dgr) && crimac_upconv_wch_hp_ok(ntaps, in_bytes, Cin, N, cout_up, out_ld),
                   "igemm (plane pairs): only ConvTranspose2d k2 s2 forward (plane-pair output) and its input gradient "
                   "(fp32 output), Cin %% 32 == 0, N %% 128 == 0, tensors below 2 GiB");
    return crimac_upconv_wch_hp(ntaps, in, in_ld, B, Ho, Wo, Cin, N, w_hi, bias, cout_up, out, out_ld, st, out_planes);
  }
  if (is16 && !relu) {
    // ConvTranspose2d(k2, s2): forward (1 tap, scatter) and input gradient (4 taps, stride 2) go to upconv.hip
    static const int use_wch = getenv("CRIMAC_UPCONV_WCH") ? atoi(getenv("CRIMAC_UPCONV_WCH")) : 1;
    const long in_bytes = (((long)B * Hi * Wi - 1) * in_ld + Cin) * 2;
    const bool fwd = out_mode == 1 && ntaps == 1 && stride == 1 && pad == 0 && Hi == Ho && Wi == Wo &&
                     (!bias || bias_mod == cout_up);
    const bool dgr = out_mode == 0 && ntaps == 4 && tw == 2 && stride == 2 && pad == 0 && Hi == 2 * Ho &&
                     Wi == 2 * Wo && !bias;
    if (use_wch && (fwd || dgr) && crimac_upconv_wch_ok(ntaps, in_bytes, Cin, N, cout_up, out_ld))
      return crimac_upconv_wch_16(ntaps, in, in_ld, B, Ho, Wo, Cin, N, w_hi, bias, cout_up, out, out_ld, st,
                                  prec == CRIMAC_PREC_FP16);
  }
  if (prec == CRIMAC_PREC_BF16)
    return out_mode == 0 ? dispatch<bf16_t, 1, 0>(p, st) : dispatch<bf16_t, 1, 1>(p, st);
  if (prec == CRIMAC_PREC_FP16)
    return out_mode == 0 ? dispatch<half_t, 1, 0>(p, st) : dispatch<half_t, 1, 1>(p, st);
  if (prec == CRIMAC_PREC_F32H3)
    return out_mode == 0 ? dispatch<float, 2, 0, half_t>(p, st) : dispatch<float, 2, 1, half_t>(p, st);
  if (prec == CRIMAC_PREC_F32X3)
    return out_mode == 0 ? dispatch<float, 2, 0>(p, st) : dispatch<float, 2, 1>(p, st);
  return out_mode == 0 ? dispatch<float, 3, 0>(p, st) : dispatch<float, 3, 1>(p, st);
}
