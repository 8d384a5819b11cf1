// Weight-gradient contraction on MFMA for CDNA4 (gfx950), NHWC activations.
//
// Reference op: the weight half of aten::convolution_backward for nn.Conv2d 3x3 (unet.py:35-44)
// and nn.ConvTranspose2d k2 s2 (unet.py:47-49) -- 64 % of the reference's CPU train step
// (SURVEY.md §3.1).
//
//   dW[t][f][s] = sum over pixels p of  F[p][f] * S[shift_t(p)][s]
//
//   conv3x3 : F = dY (Cout), S = X (Cin),  9 taps, shift_t(y,x) = (y+ky-1, x+kx-1), zero outside
//   upconv  : F = X coarse (Cin), S = dY fine (Cout), 4 taps, shift_ab(y,x) = (2y+a, 2x+b)
//
// The contraction index is the pixel, which is the *strided* index of an NHWC tensor, so both MFMA
// operands are needed transposed.  The tiles are staged pixel-major in LDS exactly as they sit in
// HBM (coalesced 16-byte loads) and read with ds_read_b64_tr_b16, gfx950's transposing LDS read:
// no register or LDS transpose pass.
//
// Workgroup = 64 F-channels x 64 S-channels x ALL taps (the S halo tile is staged once and reused
// by the 9 shifted reads), 4 waves.  bf16: 2 S halves x 2 tap groups, each wave 64 F x 32 S for its 5 / 4
// taps (10 / 8 accumulators); fp32 modes: 2x2 over channels, each wave 32x32 per tap (9 accumulators).
// Pixel tiles are TR x 16 spatial patches; the pixel range is split across workgroups and partial sums are
// combined with fp32 atomics whose wave shape is two 128-byte row segments (full atomic rate).
#include "common.h"

CRIMAC_DIAG_DECLARE(crimac_diag_clock_wgrad)

namespace {

struct WgradParams {
  const void* f; long f_ld; int CF;
  const void* s; long s_ld; int CS;
  int B, Hf, Wf;
  float* dw;
  int tiles_y, tiles_x;
  long ntiles;
  int tiles_per_block;
  int nsplits;
  long partial_stride;     // 0: dw += acc by fp32 atomics; > 0: plain stores into dw + split * partial_stride
  // 16-bit kernels only: bytes per ELEMENT of each operand tensor.  2 = a 16-bit tensor; 4 (CRIMAC_PREC_H3F_BWD, the
  // activation operand) = an fp16 plane-pair tensor of which the contraction takes the hi plane: the first 16 bytes of
  // every 32-byte [8 hi | 8 lo] group -- the same lane -> 8-channel-unit map, addressed in 4-byte elements
  int f_es = 2, s_es = 2;
};

__device__ __forceinline__ int swz_tr(int row) { return ((row >> 1) & 1) << 6; }

__device__ __forceinline__ bf16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(p));
}


// ---- bf16 contraction of one staged tile, hand-placed schedule ------------------------------------------
// Wave roles (bf16 path): 2 S-channel halves x 2 TAP GROUPS; every wave covers all 64 F channels x its 32 S
// channels x its taps (conv3x3: taps 0-4 | 5-8, up-conv: 0-1 | 2-3).
// MFMA shape v_mfma_f32_16x16x32_bf16 (4 x 2 tiles per tap): K = 32 pixels = TWO tile rows; the chip holds a
// higher clock under a dense stream of this shape than with 32x32x16 (MI355X_MICROARCH.md, clocks under load
// (7); conv3x3_glds.hip measured 8-10 %).  k index -> pixel (any bijection works, both operands use the same):
// lane block g = lane / 16 holds k = 8g .. 8g+7 = tile row 2*rp + (g >> 1), x = 4*(g & 1) + {0..3} (first
// transposing read) and x + 8 (second read).  A half-wave (blocks g, g^1) therefore reads 8 CONSECUTIVE pixel
// rows of one 32-byte channel slot per cycle: conflict-free iff the slot index is XORed with TWO row bits
// (swz16: bits 1 and 2; the 128-byte row pitch contributes bit 0).
// Schedule: a row pair is N steps (one per tap of the wave's group), a step = 8 MFMAs (4 F x 2 S fragments)
// on the S fragments requested one step earlier (2-deep ring).  The F fragments are single-buffered: in the
// last step of a row pair each fragment's two MFMAs are followed by the reads of the same fragment of the
// NEXT row pair (in-order issue: the MFMAs have taken their operands), and the first step of that row pair
// releases them one by one.  LDS reads return in order, so `s_waitcnt lgkmcnt(n)`, n = reads issued after
// the ones needed, releases exactly the fragments of the next MFMAs.
// The reads are inline asm: through the intrinsic hipcc puts `s_waitcnt vmcnt(0)` in front of the first read
// of every tile row (an LDS read "may alias" the LDS-DMA of the NEXT tile, already in flight), which drained
// the prefetch before the MFMAs.
// Addresses: S row r = c + R (c compile-time, R per lane), byte = r*128 + ((slot ^ swz16(r)) << 5) + 8*pp;
// swz16(r) depends on (c + R) mod 8: 4 per-lane bases (by c & 3; c & 4 exchanges the two slots) + the immediate
// c*128 cover every S read of one 16-channel slot, `^ 32` gives the other; F rows are multiples of 8 -> one
// base, `^ (fh << 5)`.
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr16_asm(unsigned addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
}
// consumers of x must stay behind the (volatile, ordered) wait that precedes this statement
__device__ __forceinline__ void tie(bf16x4& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ int swz16(int row) { return (((row >> 1) & 1) << 6) | (((row >> 2) & 1) << 5); }

template <int MODE, int TG>
struct WgTaps {
  static constexpr int TR = MODE == 0 ? 8 : 4;
  static constexpr int NRP = TR / 2;                                             // row pairs per tile
  static constexpr int RS = MODE == 0 ? 18 : 64;                                 // S rows between tile rows
  static constexpr int T0 = MODE == 0 ? (TG == 0 ? 0 : 5) : (TG == 0 ? 0 : 2);   // first tap of the group
  static constexpr int N = MODE == 0 ? (TG == 0 ? 5 : 4) : 2;                    // taps of the group
  static constexpr int srow(int py, int t) {
    return MODE == 0 ? (py + t / 3) * 18 + t % 3 : (2 * py + (t >> 1)) * 32 + (t & 1) * 16;
  }
};
struct WgFrags {
  bf16x4 a[4][2];         // [F fragment][read]
  bf16x4 b[2][2][2];      // [step parity][S fragment][read]
};
// both reads of F fragment FH of row pair RP
template <int RP, int FH>
__device__ __forceinline__ void wg_rd_a(unsigned fv0, WgFrags& f) {
  f.a[FH][0] = lds_tr16_asm<(RP * 32) * 128>(fv0 ^ (FH << 5));
  f.a[FH][1] = lds_tr16_asm<(RP * 32 + 8) * 128>(fv0 ^ (FH << 5));
}
// S fragments of step S (row pair S / N, local tap S % N).  swz16(c + 4 + R) = swz16(c + R) ^ 32: the bases of
// the classes c & 7 >= 4 are those of c & 3 with the two 16-channel slots exchanged -> 4 registers, not 8.
// NARROW: only the first 16-channel S fragment of the wave is real (first layer: 4 input channels padded to 16)
template <int MODE, int TG, int S, bool NARROW, typename FR>
__device__ __forceinline__ void wg_rd_b(const unsigned (&sv)[4], FR& f) {
  using G = WgTaps<MODE, TG>;
  constexpr int c = G::srow(2 * (S / G::N), G::T0 + S % G::N);
  constexpr int x = (c >> 2) & 1;
  const unsigned s0 = sv[c & 3], s1 = s0 ^ 32;
  if constexpr (!NARROW || x == 0) {
    f.b[S & 1][x][0] = lds_tr16_asm<c * 128>(s0);
    f.b[S & 1][x][1] = lds_tr16_asm<(c + 8) * 128>(s0);
  }
  if constexpr (!NARROW || x == 1) {
    f.b[S & 1][x ^ 1][0] = lds_tr16_asm<c * 128>(s1);
    f.b[S & 1][x ^ 1][1] = lds_tr16_asm<(c + 8) * 128>(s1);
  }
}
// F fragment FH of step S: wait for it (first step of a row pair), 2 MFMAs, then (last step of a row pair)
// request the same fragment of the next row pair into the registers just consumed
template <typename T16, int MODE, int TG, int S, int FH, bool NARROW, typename ACC>
__device__ __forceinline__ void wg_fh(unsigned fv0, WgFrags& f, ACC& acc) {
  using G = WgTaps<MODE, TG>;
  constexpr int NSTEP = G::NRP * G::N, rp = S / G::N, L = S % G::N;
  constexpr int NSH = NARROW ? 1 : 2;
  constexpr int nb = S + 1 < NSTEP ? 2 * NSH : 0;    // S reads of the next step, issued at the top of this one
  if constexpr (L == 0) {
    if constexpr (rp > 0 || FH == 0) wait_lgkm<nb + (rp > 0 ? 2 * (3 - FH) : 0)>();
    if constexpr (FH == 0) {
#pragma unroll
      for (int sh = 0; sh < NSH; ++sh) { tie(f.b[S & 1][sh][0]); tie(f.b[S & 1][sh][1]); }
    }
    if constexpr (rp > 0) { tie(f.a[FH][0]); tie(f.a[FH][1]); }
    else if constexpr (FH == 0) {
#pragma unroll
      for (int fh = 0; fh < 4; ++fh) { tie(f.a[fh][0]); tie(f.a[fh][1]); }
    }
  } else if constexpr (FH == 0) {
    wait_lgkm<nb>();
#pragma unroll
    for (int sh = 0; sh < NSH; ++sh) { tie(f.b[S & 1][sh][0]); tie(f.b[S & 1][sh][1]); }
  }
  const bf16x8 af = __builtin_shufflevector(f.a[FH][0], f.a[FH][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int sh = 0; sh < NSH; ++sh) {
    const bf16x8 bfr = __builtin_shufflevector(f.b[S & 1][sh][0], f.b[S & 1][sh][1], 0, 1, 2, 3, 4, 5, 6, 7);
    acc[L][FH][sh] = E16<T16>::mfma16(af, bfr, acc[L][FH][sh]);
  }
  if constexpr (L == G::N - 1 && rp + 1 < G::NRP) wg_rd_a<rp + 1, FH>(fv0, f);
  if constexpr (FH + 1 < 4) wg_fh<T16, MODE, TG, S, FH + 1, NARROW>(fv0, f, acc);
}
template <typename T16, int MODE, int TG, int S, bool NARROW, typename ACC>
__device__ __forceinline__ void wg_step(unsigned fv0, const unsigned (&sv)[4], WgFrags& f, ACC& acc) {
  using G = WgTaps<MODE, TG>;
  constexpr int NSTEP = G::NRP * G::N;
  if constexpr (S + 1 < NSTEP) wg_rd_b<MODE, TG, S + 1, NARROW>(sv, f);
  wg_fh<T16, MODE, TG, S, 0, NARROW>(fv0, f, acc);
  if constexpr (S + 1 < NSTEP) wg_step<T16, MODE, TG, S + 1, NARROW>(fv0, sv, f, acc);
}

// ---- plane pairs (CRIMAC_PREC_H3P), conv3x3: 64 x 64 REAL channels per workgroup of 8 waves ------------------------
// Both operands are fp16 plane pairs (common.h hp_t): dW = Fh^T Sh + Fh^T Sl + Fl^T Sh, 3 MFMAs per fragment pair.
// A tile of 64 real channels is 256 bytes per pixel; it is staged as TWO images per operand (real channels 0-31 and
// 32-63), each with exactly the 16-bit kernel's shape -- 128-byte rows of four 32-byte slots -- so the transposing reads,
// their swizzle and the hand-placed read / MFMA cadence above carry over.  The slots of an image row are
// [ch 0-15 hi | ch 0-15 lo | ch 16-31 hi | ch 16-31 lo]: the permutation from the [8 hi | 8 lo] groups in memory rides on
// the per-lane DMA source offset (as the swizzle does).
// Why 64 x 64 and not the 16-bit tile shape reinterpreted (32 x 32 real channels): per loaded byte that form does 3/4 of
// the 16-bit kernel's MFMAs -- and the 16-bit kernel is already limited by its LDS-DMA issue -- this one 3/2.
// Wave roles (8 waves, one workgroup per CU, double-buffered tiles, one barrier per tile): S image sa x 16-channel S
// slot pair ws x tap group tg; every wave covers all 64 F channels = 8 plane fragments (image, 16 channels, plane).
// Per tap step: 4 S reads (hi, lo fragments), 12 MFMAs (hi*lo, hi*hi on the hi F fragments, lo*hi on the lo ones).
constexpr int PP_F_IMG = 128 * 128;                 // one F image: 8 x 16 pixel rows of 128 bytes
constexpr int PP_S_IMG = 184 * 128;                 // one S image: (8 + 2) x 18 = 180 halo rows, padded to 184
constexpr int PP_BUF = 2 * PP_F_IMG + 2 * PP_S_IMG;   // 79,872 bytes per tile buffer

// Slab form of the flush (partial_stride > 0): a 64 x 64 x taps accumulator tile staged in LDS as [tap][row][kFlushPitch]
// floats goes out as 16-byte stores, 16 consecutive lanes per 256-byte tile row: whole lines per wave-instruction.
constexpr int kFlushPitch = 68;        // 64 + 4: the MFMA layout's lane blocks (4 rows apart) land 16 banks apart
template <int NTHR>
__device__ __forceinline__ void flush_staged_tile(const float* stg, float* slab, int ntaps, int CF, int CS, int cf0, int cs0,
                                                  int t_id) {
  for (int i = t_id; i < ntaps * 64 * 16; i += NTHR) {
    const int R = i >> 4, c4 = i & 15, t = R >> 6, row = R & 63;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + R * kFlushPitch + c4 * 4);
    *reinterpret_cast<f32x4*>(slab + ((long)t * CF + cf0 + row) * CS + cs0 + c4 * 4) = v;
  }
}

struct PpFrags {
  bf16x4 a[8][2];         // [F plane fragment: image * 4 + slot][read]
  bf16x4 b[2][2][2];      // [step parity][S fragment: hi, lo][read]
};
// NFW: F plane fragments per wave -- 8: all 64 F channels (fragment FH = image FH / 4, slot FH % 4); 2: the hi / lo
// planes of ONE 16-channel fragment whose image and slot pair are folded into fv0 (the first-layer form below)
template <int RP, int FH, int NFW = 8>
__device__ __forceinline__ void pp_rd_a(unsigned fv0, PpFrags& f) {
  constexpr int img_off = NFW == 8 ? (FH >> 2) * PP_F_IMG : 0;
  f.a[FH][0] = lds_tr16_asm<(RP * 32) * 128 + img_off>(fv0 ^ ((FH & 3) << 5));
  f.a[FH][1] = lds_tr16_asm<(RP * 32 + 8) * 128 + img_off>(fv0 ^ ((FH & 3) << 5));
}
// F plane fragment FH of step S (the 16-bit kernel's wg_fh with 8 fragments; even FH = hi plane: two MFMAs, with the lo
// and the hi S fragment; odd FH = lo plane: one, with the hi S fragment)
template <int TG, int S, int FH, int NFW = 8, typename ACC>
__device__ __forceinline__ void pp_fh(unsigned fv0, PpFrags& f, ACC& acc) {
  using G = WgTaps<0, TG>;
  constexpr int NSTEP = G::NRP * G::N, rp = S / G::N, L = S % G::N;
  constexpr int nb = S + 1 < NSTEP ? 4 : 0;          // S reads of the next step, issued at the top of this one
  if constexpr (L == 0) {
    // (lgkmcnt is a 4-bit counter: where more than 15 reads would be allowed to stay in flight, waiting for 15 is the
    // conservative form -- it only asks for a few reads more than needed to have landed)
    constexpr int allow = nb + (rp > 0 ? 2 * (NFW - 1 - FH) : 0);
    if constexpr (rp > 0 || FH == 0) wait_lgkm<(allow < 15 ? allow : 15)>();
    if constexpr (FH == 0) {
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) { tie(f.b[S & 1][sh][0]); tie(f.b[S & 1][sh][1]); }
    }
    if constexpr (rp > 0) { tie(f.a[FH][0]); tie(f.a[FH][1]); }
    else if constexpr (FH == 0) {
#pragma unroll
      for (int fh = 0; fh < NFW; ++fh) { tie(f.a[fh][0]); tie(f.a[fh][1]); }
    }
  } else if constexpr (FH == 0) {
    wait_lgkm<nb>();
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) { tie(f.b[S & 1][sh][0]); tie(f.b[S & 1][sh][1]); }
  }
  const bf16x8 af = __builtin_shufflevector(f.a[FH][0], f.a[FH][1], 0, 1, 2, 3, 4, 5, 6, 7);
  const bf16x8 b_hi = __builtin_shufflevector(f.b[S & 1][0][0], f.b[S & 1][0][1], 0, 1, 2, 3, 4, 5, 6, 7);
  if constexpr ((FH & 1) == 0) {
    const bf16x8 b_lo = __builtin_shufflevector(f.b[S & 1][1][0], f.b[S & 1][1][1], 0, 1, 2, 3, 4, 5, 6, 7);
    acc[L][FH >> 1] = E16<half_t>::mfma16(af, b_lo, acc[L][FH >> 1]);
  }
  acc[L][FH >> 1] = E16<half_t>::mfma16(af, b_hi, acc[L][FH >> 1]);
  if constexpr (L == G::N - 1 && rp + 1 < G::NRP) pp_rd_a<rp + 1, FH, NFW>(fv0, f);
  if constexpr (FH + 1 < NFW) pp_fh<TG, S, FH + 1, NFW>(fv0, f, acc);
}
template <int TG, int S, int NFW = 8, typename ACC>
__device__ __forceinline__ void pp_step(unsigned fv0, const unsigned (&sv)[4], PpFrags& f, ACC& acc) {
  using G = WgTaps<0, TG>;
  constexpr int NSTEP = G::NRP * G::N;
  if constexpr (S + 1 < NSTEP) wg_rd_b<0, TG, S + 1, false>(sv, f);
  pp_fh<TG, S, 0, NFW>(fv0, f, acc);
  if constexpr (S + 1 < NSTEP) pp_step<TG, S + 1, NFW>(fv0, sv, f, acc);
}

// NARROW: the first layer (S = the network input, 4 channels padded to 16: ONE 16-channel S fragment).  All 64 F channels
// x 16 S channels x 9 taps per workgroup; the 8 waves split tap group x F image x F 16-channel fragment (2 plane fragments
// and 3 MFMAs per tap step each): the launch is its F stream (dY: 256 bytes per pixel).
template <bool NARROW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_pp_kernel(WgradParams p) {
  constexpr int TR = 8;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NSI = NARROW ? 1 : 2;                  // S images staged
  constexpr int NDMA = 2 * 16 + NSI * 23;              // DMA wave-instructions per tile
  // Who moves the tiles.  All 8 waves (DW = 8): after the barrier both waves of a SIMD issue their DMA share and THEN
  // both contract -- the SIMD's MFMA pipe idles through the issue phase.  Only waves 4-7 (DW = 4, the tap group with 4 of
  // the 9 taps): while they issue the whole tile's DMA their SIMD partners (waves 0-3, 5 taps) already contract.
#ifndef CRIMAC_WGRAD_PP_DMA_WAVES
#define CRIMAC_WGRAD_PP_DMA_WAVES 4
#endif
  constexpr int DW = NARROW ? 8 : CRIMAC_WGRAD_PP_DMA_WAVES;
  constexpr int NI = (NDMA + DW - 1) / DW;             // ... per moving wave
  constexpr int NFW = NARROW ? 2 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // roles: S image sa, S slot pair ws (NARROW: the one real S fragment for everybody, F image fi / fragment ff instead)
  const int ws = NARROW ? 0 : (wave & 1), sa = NARROW ? 0 : ((wave >> 1) & 1), tg = wave >> 2;
  const int ff = wave & 1, fi = (wave >> 1) & 1;
  // workgroup -> (64 x 64 channel tile, pixel split): as the 16-bit kernel (XCD-aware: equal id % 8 share an L2)
  const int cs_tiles = NARROW ? 1 : p.CS / 64;
  const int ch_tiles = (p.CF / 64) * cs_tiles;
  int qt, split;
  if (p.nsplits % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qt = j % ch_tiles;
    split = xcd + 8 * (j / ch_tiles);
  } else if (ch_tiles % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, R = ch_tiles >> 3;
    qt = xcd * R + j % R;
    split = j / R;
  } else {
    qt = blockIdx.x % ch_tiles;
    split = blockIdx.x / ch_tiles;
  }
  const int cf0 = (qt / cs_tiles) * 64, cs0 = (qt % cs_tiles) * 64;
  const hp_t* fp = reinterpret_cast<const hp_t*>(p.f);
  const hp_t* sp = reinterpret_cast<const hp_t*>(p.s);

  // ---- DMA plan: wave-instruction k = wave + 8 i moves 8 pixel rows (1 KiB) of one image ------------------------
  // k in [0,16): F image 0 | [16,32): F image 1 | [32,55): S image 0 | [55,78): S image 1
  const int sub = lane >> 3, c = lane & 7;
  unsigned rel[NI];          // byte offset from the tile origin pixel (S: the halo's first pixel); OOB = no lane data
  int lds_at[NI];            // byte offset of the instruction's 1 KiB inside a tile buffer; -1: no such instruction
  auto geo = [&](int k, bool& is_s, int& img, int& row) {
    is_s = k >= 32;
    const int kk = is_s ? k - 32 : k;
    img = is_s ? (kk >= 23 ? 1 : 0) : (kk >> 4);
    row = 8 * (is_s ? kk - 23 * img : (kk & 15)) + sub;
  };
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = (wave & (DW - 1)) + DW * i;
    bool is_s; int img, row;
    geo(k, is_s, img, row);
    const int u = c ^ (swz16(row) >> 4);                       // logical 16-byte unit this lane's LDS position holds
    const int su = (((u >> 2) * 2 + (u & 1)) << 1) | ((u >> 1) & 1);   // its source unit: group (2 fh + half), plane
    lds_at[i] = k < NDMA ? (is_s ? 2 * PP_F_IMG + img * PP_S_IMG + (row - sub) * 128 : img * PP_F_IMG + (row - sub) * 128) : -1;
    if (!is_s) {
      const int ry = row >> 4, rx = row & 15;
      rel[i] = (unsigned)(((ry * (long)p.Wf + rx) * p.f_ld + cf0 + 32 * img) * 4 + su * 16);
    } else {
      const int ry = (row * 3641) >> 16, rx = row - ry * 18;
      // NARROW: the pixel holds 16 channels = source units 0 .. 3 (slots 0, 1); the rest of the row is zero-filled
      const bool real = row < 180 && (!NARROW || (u >> 2) == 0);
      rel[i] = real ? (unsigned)(((ry * (long)p.Wf + rx) * p.s_ld + cs0 + 32 * img) * 4 + su * 16) : OOB;
    }
  }
  auto tile_origin = [&](long tile, long& b, int& y0, int& x0) {
    const int txi = (int)(tile % p.tiles_x);
    const long tt = tile / p.tiles_x;
    const int tyi = (int)(tt % p.tiles_y);
    b = tt / p.tiles_y;
    y0 = tyi * TR;
    x0 = txi * 16;
  };
  auto issue_tile = [&](long tile, int buf) {
    long b; int y0, x0;
    tile_origin(tile, b, y0, x0);
    unsigned char* base = smem + buf * PP_BUF;
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<hp_t*>(fp + ((b * p.Hf + y0) * (long)p.Wf + x0) * p.f_ld), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<hp_t*>(sp + ((b * p.Hf + y0 - 1) * (long)p.Wf + x0 - 1) * p.s_ld), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = (wave & (DW - 1)) + DW * i;
      if (k >= NDMA) continue;                       // wave-uniform
      bool is_s; int img, row;
      geo(k, is_s, img, row);
      if (!is_s) {
        const bool ok = (y0 + (row >> 4)) < p.Hf && (x0 + (row & 15)) < p.Wf;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                 (int)(ok ? rel[i] : OOB), 0, 0, 0);
      } else {
        const int ry = (row * 3641) >> 16, rx = row - ry * 18;
        const unsigned y = (unsigned)(y0 - 1 + ry), x = (unsigned)(x0 - 1 + rx);
        const bool ok = y < (unsigned)p.Hf && x < (unsigned)p.Wf;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                 (int)(ok ? rel[i] : OOB), 0, 0, 0);
      }
    }
  };

  const long t_begin = (long)split * p.tiles_per_block;
  const long t_end = t_begin + p.tiles_per_block < p.ntiles ? t_begin + p.tiles_per_block : p.ntiles;
  // lane roles of the transposing reads (as in the 16-bit kernel)
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  auto run = [&](auto tgc) {
    constexpr int TG = decltype(tgc)::value;
    using G = WgTaps<0, TG>;
    constexpr int NFR = NFW / 2;                       // 16-channel F fragments of this wave: 4 (1)
    f32x4 acc[G::N][NFR];
#pragma unroll
    for (int t = 0; t < G::N; ++t)
#pragma unroll
      for (int fr = 0; fr < NFR; ++fr)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][fr][r] = 0.f;
    const int RF = (g >> 1) * 16 + 4 * (g & 1) + q;
    const int RSl = (g >> 1) * G::RS + 4 * (g & 1) + q;
    const unsigned lds0 = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem));
    const bool mover = DW == 8 || wave >= 4;
    if (t_begin < t_end && mover) issue_tile(t_begin, 0);
    for (long tile = t_begin; tile < t_end; ++tile) {
      const int cur = (int)((tile - t_begin) & 1);
      __syncthreads();           // vmcnt(0) + barrier: the tile has landed for everyone, the other buffer is free
      if (tile + 1 < t_end && mover) issue_tile(tile + 1, cur ^ 1);
      const unsigned aF = lds0 + cur * PP_BUF, aS = aF + 2 * PP_F_IMG + sa * PP_S_IMG;
      // (NARROW: this wave's F image and 16-channel fragment -- slots 2 ff, 2 ff + 1 -- ride in the address)
      const unsigned fv0 = aF + (NARROW ? fi * PP_F_IMG : 0) + ((RF * 128 + 8 * pp + swz16(RF)) ^ (NARROW ? ff << 6 : 0));
      unsigned sv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) sv[k] = aS + RSl * 128 + 8 * pp + ((ws * 64) ^ swz16(k + RSl));
      PpFrags f;
      pp_rd_a<0, 0, NFW>(fv0, f); pp_rd_a<0, 1, NFW>(fv0, f);
      if constexpr (!NARROW) {
        pp_rd_a<0, 2>(fv0, f); pp_rd_a<0, 3>(fv0, f);
        pp_rd_a<0, 4>(fv0, f); pp_rd_a<0, 5>(fv0, f); pp_rd_a<0, 6>(fv0, f); pp_rd_a<0, 7>(fv0, f);
      }
      wg_rd_b<0, TG, 0, false>(sv, f);
      pp_step<TG, 0, NFW>(fv0, sv, f, acc);
    }
    if constexpr (!NARROW) {
      if (p.partial_stride > 0) {                 // slab form: through LDS, whole lines (flush_staged_tile)
        float* stg = reinterpret_cast<float*>(smem);
        __syncthreads();                          // every wave has left its tile loop: the tile buffers are dead
        float* sl = stg + (G::T0 * 64 + (lane >> 4) * 4) * kFlushPitch + sa * 32 + ws * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < G::N; ++t) {
          float* st_ = sl + t * 64 * kFlushPitch;
          asm volatile("" : "+v"(st_));
#pragma unroll
          for (int fr = 0; fr < NFR; ++fr)
#pragma unroll
            for (int r = 0; r < 4; ++r) st_[(fr * 16 + r) * kFlushPitch] = acc[t][fr][r];
        }
        __syncthreads();
        flush_staged_tile<512>(stg, p.dw + (long)split * p.partial_stride, 9, p.CF, p.CS, cf0, cs0, tid);
        return;
      }
    }
    // dw[t][cf][cs] += acc: F rows cf0 + 16 fr .. +15, S columns cs0 + 32 sa + 16 ws .. +15, this wave's taps
    float* dwp = p.dw + (p.partial_stride > 0 ? (long)split * p.partial_stride : 0);
    const int col = cs0 + sa * 32 + ws * 16 + (lane & 15);
    const int fr0 = NARROW ? 2 * fi + ff : 0;          // (NARROW: this wave's one F fragment)
#pragma unroll
    for (int t = 0; t < G::N; ++t)
#pragma unroll
      for (int fr = 0; fr < NFR; ++fr)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = cf0 + (fr0 + fr) * 16 + (lane >> 4) * 4 + r;
          float* dst = dwp + ((long)(G::T0 + t) * p.CF + row) * p.CS + col;
          if (p.partial_stride > 0) *dst = acc[t][fr][r];
          else atomicAdd(dst, acc[t][fr][r]);
        }
  };
  if (tg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// ---- plane pairs, transposed convolution 2x2 / stride 2: 128 F x 64 S REAL channels x 4 taps per workgroup of 8 waves ----
// dW[ab][ci][co] = sum_p X[p][ci] dY[(2y + a, 2x + b)][co]: F = X on the coarse grid, S = dY on the fine grid.  Four taps
// give a staged byte a quarter of the MFMAs the 3x3 kernel gets out of it, so the launch is its tile traffic: with 64 x 64
// tiles every shape of the network moves 1.34 GB through the LDS-DMA path (F is re-read CS / 64 times, S CF / 64 times);
// 128 x 64 tiles move 0.8 GB.  Tile = 2 x 16 coarse pixels (one k-step of 32 pixels): F as four 32-channel images of 32
// rows, S as two images of 128 rows ordered [fine row][column parity][x] (48 KB per buffer, two buffers); rows, slots and
// the swizzle as in wgrad_pp_kernel.  Waves = S image x S slot pair x tap group (a = 0 | a = 1); every wave holds all 128
// F channels (16 plane fragments), 2 taps x 8 accumulator tiles; per tile 40 transposing reads, one wait, 48 MFMAs.
constexpr int UP_F_IMG = 32 * 128, UP_S_IMG = 128 * 128;
constexpr int UP_BUF = 4 * UP_F_IMG + 2 * UP_S_IMG;          // 49,152 bytes per tile buffer
struct UpFrags {
  bf16x4 a[16][2];        // [F plane fragment: image * 4 + slot][read]
  bf16x4 b[2][2][2];      // [tap of the group][S fragment: hi, lo][read]
};
template <int FH>
__device__ __forceinline__ void up_rd_a(unsigned fv0, UpFrags& f) {
  f.a[FH][0] = lds_tr16_asm<(FH >> 2) * UP_F_IMG>(fv0 ^ ((FH & 3) << 5));
  f.a[FH][1] = lds_tr16_asm<(FH >> 2) * UP_F_IMG + 8 * 128>(fv0 ^ ((FH & 3) << 5));
  if constexpr (FH + 1 < 16) up_rd_a<FH + 1>(fv0, f);
}
template <int A, int L>          // tap (a = A, b = L): S rows A * 32 + L * 16 + (this lane's row)
__device__ __forceinline__ void up_rd_b(unsigned s0, UpFrags& f) {
  constexpr int c = A * 32 + L * 16;
  f.b[L][0][0] = lds_tr16_asm<c * 128>(s0);
  f.b[L][0][1] = lds_tr16_asm<(c + 8) * 128>(s0);
  f.b[L][1][0] = lds_tr16_asm<c * 128>(s0 ^ 32);
  f.b[L][1][1] = lds_tr16_asm<(c + 8) * 128>(s0 ^ 32);
}
template <int L, int FH, typename ACC>
__device__ __forceinline__ void up_mfma(UpFrags& f, ACC& acc) {
  const bf16x8 af = __builtin_shufflevector(f.a[FH][0], f.a[FH][1], 0, 1, 2, 3, 4, 5, 6, 7);
  const bf16x8 b_hi = __builtin_shufflevector(f.b[L][0][0], f.b[L][0][1], 0, 1, 2, 3, 4, 5, 6, 7);
  if constexpr ((FH & 1) == 0) {
    const bf16x8 b_lo = __builtin_shufflevector(f.b[L][1][0], f.b[L][1][1], 0, 1, 2, 3, 4, 5, 6, 7);
    acc[L][FH >> 1] = E16<half_t>::mfma16(af, b_lo, acc[L][FH >> 1]);
  }
  acc[L][FH >> 1] = E16<half_t>::mfma16(af, b_hi, acc[L][FH >> 1]);
  if constexpr (FH + 1 < 16) up_mfma<L, FH + 1>(f, acc);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_up_pp_kernel(WgradParams p) {
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NDMA = 4 * 4 + 2 * 16;                 // DMA wave-instructions (8 rows = 1 KiB each) per tile
  constexpr int NI = NDMA / 4;                         // ... per moving wave (waves 4-7, as in wgrad_pp_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = wave & 1, sa = (wave >> 1) & 1, tg = wave >> 2;
  const int cs_tiles = p.CS / 64;
  const int ch_tiles = (p.CF / 128) * cs_tiles;
  int qt, split;
  if (p.nsplits % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qt = j % ch_tiles;
    split = xcd + 8 * (j / ch_tiles);
  } else if (ch_tiles % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, R = ch_tiles >> 3;
    qt = xcd * R + j % R;
    split = j / R;
  } else {
    qt = blockIdx.x % ch_tiles;
    split = blockIdx.x / ch_tiles;
  }
  const int cf0 = (qt / cs_tiles) * 128, cs0 = (qt % cs_tiles) * 64;
  const hp_t* fp = reinterpret_cast<const hp_t*>(p.f);
  const hp_t* sp = reinterpret_cast<const hp_t*>(p.s);
  const int Ws = 2 * p.Wf;

  // DMA plan of a moving wave: instruction k = (wave & 3) + 4 i; k in [0,16): F image k / 4, rows 8 (k % 4) ..; [16,48): S image
  const int sub = lane >> 3, c = lane & 7;
  unsigned rel[NI];
  int lds_at[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = (wave & 3) + 4 * i;
    const bool is_s = k >= 16;
    const int kk = is_s ? k - 16 : k;
    const int img = is_s ? kk >> 4 : kk >> 2;
    const int row = 8 * (is_s ? (kk & 15) : (kk & 3)) + sub;
    const int u = c ^ (swz16(row) >> 4);                       // logical 16-byte unit this lane's LDS position holds
    const int su = (((u >> 2) * 2 + (u & 1)) << 1) | ((u >> 1) & 1);   // its source unit: group (2 fh + half), plane
    lds_at[i] = is_s ? 4 * UP_F_IMG + img * UP_S_IMG + (row - sub) * 128 : img * UP_F_IMG + (row - sub) * 128;
    if (!is_s) {
      rel[i] = (unsigned)((((row >> 4) * (long)p.Wf + (row & 15)) * p.f_ld + cf0 + 32 * img) * 4 + su * 16);
    } else {
      const int fy = row >> 5, b = (row >> 4) & 1, x = row & 15;            // fine pixel (fy, 2 x + b) of the tile
      rel[i] = (unsigned)(((fy * (long)Ws + 2 * x + b) * p.s_ld + cs0 + 32 * img) * 4 + su * 16);
    }
  }
  auto issue_tile = [&](long tile, int buf) {
    const int txi = (int)(tile % p.tiles_x);
    const long tt = tile / p.tiles_x;
    const int y0 = (int)(tt % p.tiles_y) * 2, x0 = txi * 16;
    const long b = tt / p.tiles_y;
    unsigned char* base = smem + buf * UP_BUF;
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<hp_t*>(fp + ((b * p.Hf + y0) * (long)p.Wf + x0) * p.f_ld), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<hp_t*>(sp + ((b * 2 * p.Hf + 2 * y0) * (long)Ws + 2 * x0) * p.s_ld), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = (wave & 3) + 4 * i;
      const bool is_s = k >= 16;
      const int kk = is_s ? k - 16 : k;
      const int row = 8 * (is_s ? (kk & 15) : (kk & 3)) + sub;
      // coarse pixel of the row: F (row >> 4, row & 15); S fine row row >> 5 -> coarse row (row >> 6), column row & 15
      const int cy = is_s ? (row >> 6) : (row >> 4), cx = row & 15;
      const bool ok = (y0 + cy) < p.Hf && (x0 + cx) < p.Wf;
      if (!is_s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                 (int)(ok ? rel[i] : OOB), 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                 (int)(ok ? rel[i] : OOB), 0, 0, 0);
    }
  };

  const long t_begin = (long)split * p.tiles_per_block;
  const long t_end = t_begin + p.tiles_per_block < p.ntiles ? t_begin + p.tiles_per_block : p.ntiles;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int RF = (g >> 1) * 16 + 4 * (g & 1) + q;        // F row of this lane's k-group (coarse row g / 2)
  const int RSl = (g >> 1) * 64 + 4 * (g & 1) + q;       // S row: coarse row y -> fine rows 2 y, 2 y + 1 = 64 rows further
  const unsigned lds0 = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem));
  const bool mover = wave >= 4;
  auto run = [&](auto tgc) {
    constexpr int A = decltype(tgc)::value;              // this wave's taps: (a = A, b = 0), (a = A, b = 1)
    f32x4 acc[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int fr = 0; fr < 8; ++fr)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][fr][r] = 0.f;
    if (t_begin < t_end && mover) issue_tile(t_begin, 0);
    for (long tile = t_begin; tile < t_end; ++tile) {
      const int cur = (int)((tile - t_begin) & 1);
      __syncthreads();           // vmcnt(0) + barrier: the tile has landed for everyone, the other buffer is free
      if (tile + 1 < t_end && mover) issue_tile(tile + 1, cur ^ 1);
      const unsigned aF = lds0 + cur * UP_BUF, aS = aF + 4 * UP_F_IMG + sa * UP_S_IMG;
      const unsigned fv0 = aF + RF * 128 + 8 * pp + swz16(RF);
      const unsigned s0 = aS + RSl * 128 + 8 * pp + ((ws * 64) ^ swz16(RSl));
      UpFrags f;
      up_rd_a<0>(fv0, f);
      up_rd_b<A, 0>(s0, f);
      up_rd_b<A, 1>(s0, f);
      wait_lgkm<0>();
#pragma unroll
      for (int fh = 0; fh < 16; ++fh) { tie(f.a[fh][0]); tie(f.a[fh][1]); }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) { tie(f.b[t][sh][0]); tie(f.b[t][sh][1]); }
      up_mfma<0, 0>(f, acc);
      up_mfma<1, 0>(f, acc);
    }
    // dw[ab][ci][co] += acc: F rows cf0 + 16 fr .. +15, S columns cs0 + 32 sa + 16 ws .. +15, taps 2 A, 2 A + 1
    float* dwp = p.dw + (p.partial_stride > 0 ? (long)split * p.partial_stride : 0);
    const int col = cs0 + sa * 32 + ws * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int fr = 0; fr < 8; ++fr)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = cf0 + fr * 16 + (lane >> 4) * 4 + r;
          float* dst = dwp + ((long)(2 * A + t) * p.CF + row) * p.CS + col;
          if (p.partial_stride > 0) *dst = acc[t][fr][r];
          else atomicAdd(dst, acc[t][fr][r]);
        }
  };
  if (tg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// MODE 0: conv3x3 (TR = 8, halo 10 x 18);  MODE 1: upconv 2x2 (TR = 4, fine patch 8 x 32)
//
// bf16 path: tiles go global -> LDS directly (LDS-DMA through a buffer resource rebased on the tile origin,
// no staging registers), double buffered: the loads of tile t+1 are in flight while tile t is contracted;
// one barrier per tile.  The LDS image of such a load is lane-linear (1 KiB = 8 pixel rows per
// wave-instruction), so the tr-read swizzle is applied on the per-lane SOURCE offset; lanes whose pixel is
// outside the image read out of range and the hardware stores zeros.  The contraction itself runs on the
// hand-placed read/MFMA schedule above (wg_row).
// Tried and rejected: 16-row tiles in ONE buffer (73 KB; half the barriers and fragment prologues per MFMA, 1.27x
// instead of 1.41x halo re-read; the two workgroups of a CU covering each other's tile loads as in the
// convolution kernels): 2.26 vs 2.23 ms over the 13 conv shapes, the 64->64 @256x256 shape 8 % slower -- the
// prefetch of the double-buffered form is worth more than the saved barriers.  (Its first build ran 6x slower:
// the tile-DMA lambda had stopped being inlined and every call went through the scratch stack.)
// Tried and rejected: warming the XCD's L2 two tiles ahead with one ordinary 4-byte load per 128-byte line (the
// LDS budget allows only one tile in flight per workgroup and the level-0 shapes wait for their tiles at
// 3.9 TB/s): every shape got 8-15 % SLOWER (2.49 vs 2.25 ms over the 13 conv shapes).
// fp32 (split-bf16) path: register staging with the hi/lo split, single buffer.
//
// TEAMS = 2 (16-bit storage, conv3x3): ONE workgroup of 8 waves per CU instead of two of 4.  Two workgroups of a CU are
// not equals -- the SIMD arbiter serves the older wave first: of 512 workgroups with 32 tiles each, the 256 dispatched
// first left their tile loop after 82 us and the others after 123 us (diag_wgrad_phases.py) -- and each ends with its
// own 147 KB of fp32 atomics.  Here the two teams of 4 waves work on the SAME output tile and pixel range: they take
// tiles from a queue in LDS (whoever is faster takes more; both leave the loop together), run on team barriers (LDS
// counters; s_barrier would put both in the same phase), and team 1's accumulators are added to team 0's through
// LDS before the atomics: half the splits, half the atomic volume per launch.
template <typename TA, int NPL, int MODE, bool NARROW = false, int TEAMS = 1>
__global__ __launch_bounds__(256 * TEAMS) __attribute__((amdgpu_waves_per_eu(2, 2)))      // <= 256 registers: 8 waves per CU
void wgrad_kernel(WgradParams p) {
  static_assert(TEAMS == 1 || (sizeof(TA) == 2 && MODE == 0), "two teams: 16-bit conv3x3 only (LDS: 4 tile buffers)");
  constexpr bool X3 = sizeof(TA) == 4;     // fp32 activations, split into NPL bf16 planes
  constexpr bool PRE = __is_same(TA, hp_t);   // ... or fp16 plane pairs split by their producer (common.h)
  static_assert(X3 ? (NPL == 2 || NPL == 3) : NPL == 1, "bf16 -> 1 plane, fp32 -> 2 or 3 planes");
  static_assert(!PRE || NPL == 2, "plane pairs are two planes");
  constexpr int TR = MODE == 0 ? 8 : 4;
  constexpr int NTAPS = MODE == 0 ? 9 : 4;
  constexpr int F_ROWS = TR * 16;
  constexpr int S_ROWS = MODE == 0 ? (TR + 2) * 18 : (2 * TR) * 32;
  constexpr int S_ROWS_PAD = (S_ROWS + 7) / 8 * 8;
  constexpr int F_BYTES = F_ROWS * 128;
  constexpr int S_BYTES = S_ROWS_PAD * 128;
  constexpr int BUF_BYTES = F_BYTES + S_BYTES;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CRIMAC_DIAG_PHASES
  unsigned long long rt[4];
#define CRIMAC_WRT(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
  CRIMAC_WRT(0)
#else
#define CRIMAC_WRT(k)
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int team = TEAMS == 2 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;       // (wave: index inside the team)
  const int wave = TEAMS == 2 ? __builtin_amdgcn_readfirstlane((tid >> 6) & 3) : tid >> 6;
  const int wf = wave >> 1, ws = wave & 1;
  // Workgroup -> (channel-tile pair q, pixel split).  Workgroups with equal id % 8 share an XCD (and
  // its 4 MiB L2): give each XCD a contiguous block of channel-tile pairs (same F channels, a run of
  // S channels) and let it walk the splits, so the F/S pixel ranges it streams are shared by all its
  // workgroups instead of every XCD pulling every tensor through its own L2.
  const int cs_tiles = (p.CS + 63) / 64;
  const int ch_tiles = ((p.CF + 63) / 64) * cs_tiles;
  int qt, split;
  if (p.nsplits % 8 == 0) {
    // split-major: XCD x owns the pixel ranges x, x+8, ... and runs ALL channel-tile pairs on them, so a
    // staged F / S tile is fetched into that XCD's L2 once and hit by the other pairs (measured +2 %)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qt = j % ch_tiles;
    split = xcd + 8 * (j / ch_tiles);
  } else if (ch_tiles % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, R = ch_tiles >> 3;
    qt = xcd * R + j % R;
    split = j / R;
  } else {
    qt = blockIdx.x % ch_tiles;
    split = blockIdx.x / ch_tiles;
  }
  const int cf0 = (qt / cs_tiles) * 64;
  const int cs0 = (qt % cs_tiles) * 64;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const TA* fp = reinterpret_cast<const TA*>(p.f);
  const TA* sp = reinterpret_cast<const TA*>(p.s);
  const int Hs = MODE == 0 ? p.Hf : 2 * p.Hf, Ws = MODE == 0 ? p.Wf : 2 * p.Wf;

  // lane roles for the transposing reads
  const int g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  const int kh = g >> 1, cgrp = g & 1;
  const int f_chb = (wf * 32 + 16 * cgrp + 4 * pp) * 2;   // byte offset of this lane's 4 channels
  const int s_chb = (ws * 32 + 16 * cgrp + 4 * pp) * 2;

  // contraction over one staged tile: one k16 step per tile row.  Plane k of F at sF + k*F_BYTES,
  // plane k of S at sS + k*S_BYTES.
  auto contract = [&](const unsigned char* sF, const unsigned char* sS) {
#pragma unroll 1
    for (int py = 0; py < TR; ++py) {
      bf16x8 af[NPL];
      {
        const int r0 = py * 16 + 8 * kh + q, r1 = r0 + 4;
        const int o0 = r0 * 128 + (f_chb ^ swz_tr(r0)), o1 = r1 * 128 + (f_chb ^ swz_tr(r1));
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          bf16x4 x0v = lds_tr16(sF + k * F_BYTES + o0), x1v = lds_tr16(sF + k * F_BYTES + o1);
          af[k] = __builtin_shufflevector(x0v, x1v, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        int r0;
        if constexpr (MODE == 0) {
          r0 = (py + t / 3) * 18 + (t % 3) + 8 * kh + q;
        } else {
          r0 = (2 * py + (t >> 1)) * 32 + (t & 1) * 16 + 8 * kh + q;
        }
        const int r1 = r0 + 4;
        const int o0 = r0 * 128 + (s_chb ^ swz_tr(r0)), o1 = r1 * 128 + (s_chb ^ swz_tr(r1));
        bf16x8 bfr[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          bf16x4 x0v = lds_tr16(sS + k * S_BYTES + o0), x1v = lds_tr16(sS + k * S_BYTES + o1);
          bfr[k] = __builtin_shufflevector(x0v, x1v, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        mfma_planes<NPL, typename PlaneOf<TA>::type>(af, bfr, acc[t]);
      }
    }
  };

  // pixel of S image row `row` for the tile at (y0, x0)
  auto s_pixel = [&](int row, int y0, int x0, int& y, int& x) {
    if constexpr (MODE == 0) {
      y = y0 + row / 18 - 1;
      x = x0 + row % 18 - 1;
    } else {
      const int fy = row >> 5, rem = row & 31;
      y = 2 * y0 + fy;
      x = 2 * (x0 + (rem & 15)) + (rem >> 4);
    }
  };
  auto tile_origin = [&](long tile, long& b, int& y0, int& x0) {
    const int txi = (int)(tile % p.tiles_x);
    const long tt = tile / p.tiles_x;
    const int tyi = (int)(tt % p.tiles_y);
    b = tt / p.tiles_y;
    y0 = tyi * TR;
    x0 = txi * 16;
  };

  const long t_begin = (long)split * p.tiles_per_block;
  const long t_end = t_begin + p.tiles_per_block < p.ntiles ? t_begin + p.tiles_per_block : p.ntiles;

  if constexpr (!X3) {
    // ---- bf16: direct-to-LDS double buffering ----------------------------------------------------
    // Tiles come in through buffer resources rebased on the tile origin (scalar work): a lane only keeps its
    // 32-bit offset relative to that origin, and a lane whose pixel is outside the image (or whose channels
    // are past the tensor's) uses an out-of-range offset -- the hardware then writes ZEROS to its LDS slot
    // (checked on MI355X), so there is no masked-lane branch, no zero-store path and every wave issues the
    // same number of DMAs for every tile.
    constexpr int NF = F_ROWS / 8 / 4;                       // F wave-instructions per wave
    constexpr int NS = (S_ROWS_PAD / 8 + 3) / 4;             // S wave-instructions per wave
    constexpr unsigned OOB = 0x80000000u;                    // >= num_records of the rebased resources
    const int sub = lane >> 3, c = lane & 7;
    unsigned f_rel[NF], s_rel[NS];    // byte offset from the tile origin pixel (S: origin of the halo / fine patch)
    auto f_geo = [&](int i, int& ry, int& rx) {
      const int row = 8 * (wave + 4 * i) + sub;
      ry = row >> 4;
      rx = row & 15;
    };
    auto s_geo = [&](int i, int& ry, int& rx) {   // relative to (y0 - 1, x0 - 1) for MODE 0, to (2*y0, 2*x0) for MODE 1
      const int row = 8 * (wave + 4 * i) + sub;
      if constexpr (MODE == 0) {
        ry = (row * 3641) >> 16;      // row / 18 for row < 1024
        rx = row - ry * 18;
      } else {
        ry = row >> 5;
        rx = 2 * (row & 15) + ((row & 31) >> 4);
      }
    };
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int row = 8 * (wave + 4 * i) + sub;
      const int u = c ^ (swz16(row) >> 4);
      int ry, rx;
      f_geo(i, ry, rx);
      f_rel[i] = (cf0 + u * 8) < p.CF ? (unsigned)(((ry * (long)p.Wf + rx) * p.f_ld + cf0 + u * 8) * p.f_es) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int k = wave + 4 * i;
      const int row = 8 * k + sub;
      const int u = c ^ (swz16(row) >> 4);
      int ry, rx;
      s_geo(i, ry, rx);
      const bool ok = k < S_ROWS_PAD / 8 && row < S_ROWS && (cs0 + u * 8) < p.CS;
      s_rel[i] = ok ? (unsigned)(((ry * (long)Ws + rx) * p.s_ld + cs0 + u * 8) * p.s_es) : OOB;
    }
    // an operand tile that only ONE channel tile of the other operand multiplies is read exactly once from memory:
    // non-temporal (it would only displace what other kernels keep in the caches); otherwise the default policy --
    // the workgroups of the other channel tiles find it in L2
#ifdef CRIMAC_EXP_WGRAD_NT
    const bool stream_f = cs_tiles == 1, stream_s = (p.CF + 63) / 64 == 1;
#else
    const bool stream_f = false, stream_s = false;
#endif
    auto issue_tile = [&](long tile, int buf) {
      long b; int y0, x0;
      tile_origin(tile, b, y0, x0);
      unsigned char* base = smem + (team * 2 + buf) * BUF_BYTES;
      const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(fp) + ((b * p.Hf + y0) * (long)p.Wf + x0) * p.f_ld * p.f_es), 0,
          0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        int ry, rx;
        f_geo(i, ry, rx);
        const bool ok = (y0 + ry) < p.Hf && (x0 + rx) < p.Wf;
        if (stream_f)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024),
                                                   16, (int)(ok ? f_rel[i] : OOB), 0, 0, 2);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024),
                                                   16, (int)(ok ? f_rel[i] : OOB), 0, 0, 0);
      }
      unsigned char* sbase = base + F_BYTES;
      const int sy0 = MODE == 0 ? y0 - 1 : 2 * y0, sx0 = MODE == 0 ? x0 - 1 : 2 * x0;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(sp) + ((b * Hs + sy0) * (long)Ws + sx0) * p.s_ld * p.s_es), 0,
          0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int k = wave + 4 * i;
        if (k >= S_ROWS_PAD / 8) continue;          // wave-uniform
        int ry, rx;
        s_geo(i, ry, rx);
        const unsigned y = (unsigned)(sy0 + ry), x = (unsigned)(sx0 + rx);
        const bool ok = y < (unsigned)Hs && x < (unsigned)Ws;
        if (stream_s)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sbase + k * 1024), 16,
                                                   (int)(ok ? s_rel[i] : OOB), 0, 0, 2);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sbase + k * 1024), 16,
                                                   (int)(ok ? s_rel[i] : OOB), 0, 0, 0);
      }
    };
    // wave -> (S half ws, tap group tg); the tile loop and the epilogue are instantiated per tap group
    const int tg = wave >> 1;
    auto run = [&](auto tgc) {
      constexpr int TG = decltype(tgc)::value;
      using G = WgTaps<MODE, TG>;
      f32x4 acc2[G::N][4][2];
#pragma unroll
      for (int t = 0; t < G::N; ++t)
#pragma unroll
        for (int fh = 0; fh < 4; ++fh)
#pragma unroll
          for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[t][fh][sh][r] = 0.f;
      // lane's k block g = lane / 16: tile row (g >> 1) of the row pair, x = 4 * (g & 1) + q (+ 8: second read)
      const int RF = (g >> 1) * 16 + 4 * (g & 1) + q;
      const int RSl = (g >> 1) * G::RS + 4 * (g & 1) + q;
      const unsigned lds0 = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem)) + team * 2 * BUF_BYTES;
      auto contract_tile = [&](int cur) {
#ifndef CRIMAC_EXP_NOCOMPUTE
        const unsigned aF = lds0 + cur * BUF_BYTES, aS = aF + F_BYTES;
        const unsigned fv0 = aF + RF * 128 + 8 * pp + swz16(RF);
        unsigned sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[k] = aS + RSl * 128 + 8 * pp + ((ws * 64) ^ swz16(k + RSl));
        WgFrags f;
        // NARROW (first layer, CS <= 16): only S fragment 0 of the ws = 0 waves is real; the ws = 1 waves just
        // move tiles
        if (!NARROW || ws == 0) {
          wg_rd_a<0, 0>(fv0, f);
          wg_rd_a<0, 1>(fv0, f);
          wg_rd_a<0, 2>(fv0, f);
          wg_rd_a<0, 3>(fv0, f);
          wg_rd_b<MODE, TG, 0, NARROW>(sv, f);
          wg_step<typename PlaneOf<TA>::type, MODE, TG, 0, NARROW>(fv0, sv, f, acc2);
        }
#endif
      };
#ifdef CRIMAC_DIAG_PHASES
      unsigned long long ph[3] = {0, 0, 0}, ph_t;
#define CRIMAC_WPH(k) { unsigned long long tn; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[k] += tn - ph_t; ph_t = tn; }
#else
#define CRIMAC_WPH(k)
#endif
      long ntaken = 0;                           // (diagnostics: tiles this team contracted)
      if constexpr (TEAMS == 2) {
        // tile queue of the workgroup: tq[2] = next tile (relative to t_begin); tq[4 + 2 * team + parity] = the tile the
        // team requests in its NEXT iteration, taken by its first lane one iteration ahead
        unsigned* tq = reinterpret_cast<unsigned*>(smem + 4 * BUF_BYTES);
        if (tid < 3) tq[tid] = tid == 2 ? 2u : 0u;
        __syncthreads();
        const int tt = tid & 255;
        unsigned* tcnt = tq + team;
        unsigned ttarget = 0;
        long cur_tile = t_begin + team;
        bool have = cur_tile < t_end;
        if (have) issue_tile(cur_tile, 0);
        // (partial slabs promise bit-reproducible sums: there the teams take alternate tiles, whoever is faster)
        const bool fixed_order = p.partial_stride > 0;
        unsigned my_next = 2u + team;
        auto take = [&]() -> unsigned {
          if (fixed_order) { const unsigned v = my_next; my_next += 2; return v; }
          return __hip_atomic_fetch_add(tq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (tt == 0) tq[4 + 2 * team] = take();
        CRIMAC_DIAG_STAMP(dg_t0, dg_r0)
        CRIMAC_WRT(1)
#ifdef CRIMAC_DIAG_PHASES
        ph_t = dg_t0;
#endif
        int cur = 0, par = 0;
        while (have) {
          // this wave's share of the tile has landed; then the team: all of it has, and the other buffer is free
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          ttarget += 4;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          while (__hip_atomic_load(tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ttarget) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");
          CRIMAC_WPH(0)
          const long nxt = t_begin + __builtin_amdgcn_readfirstlane((int)*(volatile unsigned*)(tq + 4 + 2 * team + par));
          par ^= 1;
          if (tt == 0) tq[4 + 2 * team + par] = take();
          const bool has_next = nxt < t_end;
#ifndef CRIMAC_EXP_NOLOAD
          if (has_next) issue_tile(nxt, cur ^ 1);
#endif
          CRIMAC_WPH(1)
          contract_tile(cur);
          CRIMAC_WPH(2)
          cur ^= 1;
          cur_tile = nxt;
          have = has_next;
          ++ntaken;
        }
#ifndef CRIMAC_DIAG_PHASES
        CRIMAC_DIAG_STAMP(dg_t1, dg_r1)
        if (team == 0) { CRIMAC_DIAG_STORE(crimac_diag_clock_wgrad, dg_t0, dg_r0, dg_t1, dg_r1) }
#endif
      } else {
        if (t_begin < t_end) issue_tile(t_begin, 0);
        CRIMAC_DIAG_STAMP(dg_t0, dg_r0)
        CRIMAC_WRT(1)
#ifdef CRIMAC_DIAG_PHASES
        ph_t = dg_t0;
#endif
        for (long tile = t_begin; tile < t_end; ++tile) {
          const int cur = (int)((tile - t_begin) & 1);
          __syncthreads();     // vmcnt(0)+barrier: tile landed for everyone; the other buffer is free
          CRIMAC_WPH(0)
#ifndef CRIMAC_EXP_NOLOAD
          if (tile + 1 < t_end) issue_tile(tile + 1, cur ^ 1);
#endif
          CRIMAC_WPH(1)
          contract_tile(cur);
          CRIMAC_WPH(2)
          ++ntaken;
        }
#ifndef CRIMAC_DIAG_PHASES
        CRIMAC_DIAG_STAMP(dg_t1, dg_r1)
        CRIMAC_DIAG_STORE(crimac_diag_clock_wgrad, dg_t0, dg_r0, dg_t1, dg_r1)
#endif
      }
      CRIMAC_WRT(2)
#ifdef CRIMAC_DIAG_PHASES
      // cycles per phase of waves 0 (tap group 0) and 2 (tap group 1) of each team: wait for the tile | DMA issue |
      // contraction; slot = (workgroup, team, tap group)
      if (lane == 0 && (wave & 1) == 0 && blockIdx.x < 256)
        for (int k = 0; k < 3; ++k) {
          crimac_diag_clock_wgrad_buf[2 * (((blockIdx.x * 2 + team) * 2 + (wave >> 1)) * 3 + k)] = ph[k];
          crimac_diag_clock_wgrad_buf[2 * (((blockIdx.x * 2 + team) * 2 + (wave >> 1)) * 3 + k) + 1] = (unsigned long long)ntaken;
        }
#endif
      if constexpr (TEAMS == 2) {
        // team 1's accumulators -> LDS (the tile buffers are dead once every wave has left its loop) -> added to team 0's
        constexpr int N0 = WgTaps<MODE, 0>::N, N1 = WgTaps<MODE, 1>::N;
        static_assert((2 * N0 + 2 * N1) * 8192 <= 4 * BUF_BYTES, "accumulator exchange area");
        unsigned char* xa = smem + (TG == 0 ? ws * N0 : 2 * N0 + ws * N1) * 8192 + lane * 16;
        __syncthreads();
        if (team == 1) {
#pragma unroll
          for (int t = 0; t < G::N; ++t)
#pragma unroll
            for (int fh = 0; fh < 4; ++fh)
#pragma unroll
              for (int sh = 0; sh < 2; ++sh) *reinterpret_cast<f32x4*>(xa + ((t * 4 + fh) * 2 + sh) * 1024) = acc2[t][fh][sh];
        }
        __syncthreads();
        if (team == 1) return;
#pragma unroll
        for (int t = 0; t < G::N; ++t)
#pragma unroll
          for (int fh = 0; fh < 4; ++fh)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(xa + ((t * 4 + fh) * 2 + sh) * 1024);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc2[t][fh][sh][r] += o[r];
            }
      }
      if constexpr (TEAMS == 2) {
        if (p.partial_stride > 0 && cf0 + 64 <= p.CF && cs0 + 64 <= p.CS) {      // (workgroup-uniform)
          // slab form, whole tile: accumulators -> LDS [tap][row][68] -> 16-byte stores, 16 lanes per 256-byte tile row
          // (straight from the MFMA layout a wave-instruction writes four 64-byte half lines)
          float* stg = reinterpret_cast<float*>(smem);
          __syncthreads();                          // the exchange area above has been read
          // (one base per tap + immediates: the offsets of a tap fit the 16-bit immediate of ds_write)
          float* sl = stg + (G::T0 * 64 + (lane >> 4) * 4) * kFlushPitch + ws * 32 + (lane & 15);
#pragma unroll
          for (int t = 0; t < G::N; ++t) {
            float* st_ = sl + t * 64 * kFlushPitch;
            asm volatile("" : "+v"(st_));
#pragma unroll
            for (int fh = 0; fh < 4; ++fh)
#pragma unroll
              for (int sh = 0; sh < 2; ++sh)
#pragma unroll
                for (int r = 0; r < 4; ++r) st_[(fh * 16 + r) * kFlushPitch + sh * 16] = acc2[t][fh][sh][r];
          }
          __syncthreads();
          flush_staged_tile<256>(stg, p.dw + (long)split * p.partial_stride, NTAPS, p.CF, p.CS, cf0, cs0, tid & 255);
          return;
        }
      }
      // dw[t][cf][cs] += acc: this wave holds F rows cf0 .. cf0+63 x S columns cs0 + 32*ws .. +31 of its taps.
      // partial_stride > 0: every (channel tile, split) workgroup OWNS its tile of slab `split` -- plain stores, no
      // zero fill, no atomics (each split used to cost an fp32-atomic pass over dW at the chip-wide atomic rate of
      // ~1.3 TB/s, issued by all workgroups of the single resident round at once: an unoverlapped tail of ~50 us per
      // launch); the slabs are added up, in a fixed order, by crimac_unpack_wgrad_layers.
      float* dwp = p.dw + (p.partial_stride > 0 ? (long)split * p.partial_stride : 0);
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        const int col = cs0 + ws * 32 + sh * 16 + (lane & 15);
#ifdef CRIMAC_EXP_NOATOMIC
        if (col < 0) {
#else
        if (col < p.CS) {
#endif
#pragma unroll
          for (int t = 0; t < G::N; ++t)
#pragma unroll
            for (int fh = 0; fh < 4; ++fh)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = cf0 + fh * 16 + (lane >> 4) * 4 + r;
                if (row < p.CF) {
                  float* dst = dwp + ((long)(G::T0 + t) * p.CF + row) * p.CS + col;
                  if (p.partial_stride > 0) *dst = acc2[t][fh][sh][r];
                  else atomicAdd(dst, acc2[t][fh][sh][r]);
                }
              }
        }
      }
    };
    if (tg == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
#ifdef CRIMAC_DIAG_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the atomics issued -- not their completion)
    CRIMAC_WRT(3)
    if (lane == 0 && wave == 0 && team == 0 && blockIdx.x < 512)
      for (int k = 0; k < 4; ++k) crimac_diag_clock_wgrad_buf[6144 + blockIdx.x * 4 + k] = rt[k];
#endif
    return;
  } else {
    // ---- fp32 activations: register staging with the plane split ------------------------------------
    unsigned char* sF = smem;                       // NPL planes of F
    unsigned char* sS = smem + NPL * F_BYTES;       // NPL planes of S
    auto stage_unit = [&](const TA* src, bool ok, unsigned char* img, int plane_bytes, int o) {
      u32x4 v0 = u32x4{0, 0, 0, 0}, v1 = u32x4{0, 0, 0, 0};
      if (ok) {
        v0 = *reinterpret_cast<const u32x4*>(src);
        v1 = *reinterpret_cast<const u32x4*>(src + 4);
      }
      if constexpr (PRE) {                      // [8 hi | 8 lo] as stored: the two planes of this 8-channel group
        *reinterpret_cast<u32x4*>(img + o) = v0;
        *reinterpret_cast<u32x4*>(img + plane_bytes + o) = v1;
      } else {
        u32x4 pl[NPL];
        split8<NPL>(v0, v1, pl);
#pragma unroll
        for (int k = 0; k < NPL; ++k) *reinterpret_cast<u32x4*>(img + k * plane_bytes + o) = pl[k];
      }
    };
    for (long tile = t_begin; tile < t_end; ++tile) {
      long b; int y0, x0;
      tile_origin(tile, b, y0, x0);
      __syncthreads();   // previous tile's fragment reads are done
      for (int uidx = tid; uidx < F_ROWS * 8; uidx += 256) {
        const int row = uidx >> 3, u = uidx & 7;
        const int y = y0 + (row >> 4), x = x0 + (row & 15);
        const bool ok = y < p.Hf && x < p.Wf && (cf0 + u * 8) < p.CF;
        const TA* src = fp + ((b * p.Hf + y) * (long)p.Wf + x) * p.f_ld + cf0 + u * 8;
        stage_unit(src, ok, sF, F_BYTES, row * 128 + ((u * 16) ^ swz_tr(row)));
      }
      for (int uidx = tid; uidx < S_ROWS * 8; uidx += 256) {
        const int row = uidx >> 3, u = uidx & 7;
        int y, x;
        s_pixel(row, y0, x0, y, x);
        const bool ok = y >= 0 && y < Hs && x >= 0 && x < Ws && (cs0 + u * 8) < p.CS;
        const TA* src = sp + ((b * Hs + y) * (long)Ws + x) * p.s_ld + cs0 + u * 8;
        stage_unit(src, ok, sS, S_BYTES, row * 128 + ((u * 16) ^ swz_tr(row)));
      }
      __syncthreads();
      contract(sF, sS);
    }
  }

  // ---- combine partial sums: dw[t][cf][cs] += acc ---------------------------------------------
  const int col = cs0 + ws * 32 + (lane & 31);
#ifdef CRIMAC_EXP_NOATOMIC
  if (col < 0) {
#else
  if (col < p.CS) {
#endif
    float* dwp = p.dw + (p.partial_stride > 0 ? (long)split * p.partial_stride : 0);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cf0 + wf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.CF) {
          float* dst = dwp + ((long)t * p.CF + row) * p.CS + col;
          if (p.partial_stride > 0) *dst = acc[t][r];
          else atomicAdd(dst, acc[t][r]);
        }
      }
    }
  }
}

// ---- grouped launch: ONE persistent kernel for the conv3x3 weight gradients of several layers ---------------------------
// Every launch of wgrad_kernel ends with all 256 workgroups flushing their 147 KB of accumulators by fp32 atomics AT THE
// SAME TIME: 37.5 MB at the memory-side atomic rate (~1.2 TB/s whoever shares an address) = ~30 us in which no MFMA
// runs, 22 times per step (skipping the flush in an ablation build took 0.44 ms off the 12.0 ms bf16 step).  Here the work
// of several layers -- items (layer, 64 x 64 channel tile, pixel split) -- goes through per-XCD queues that persistent
// workgroups (one of two teams per CU, as in wgrad_kernel<.., TEAMS = 2>) pull from: a workgroup that has finished an
// item issues its atomics and goes on to the next item, whose tiles team 1 is already contracting, so the flushes of the
// 256 workgroups drift apart and drain under the other workgroups' MFMAs; only the last items' flushes are a tail.
// Since the queue balances the load, a layer no longer needs exactly one resident round of splits: the plan uses
// ~half as many (half the atomic volume per step).  16-bit storage, conv3x3, Cin >= 64 only (the first layer and the
// transposed convolutions keep their own launches).
constexpr int kGroupMaxLayers = CRIMAC_WGRAD_GROUP_MAX_LAYERS;
using GroupLayer = crimac_wgrad_group_layer;       // (include/crimac_unet_hip.h)
struct GroupParams {
  GroupLayer layer[kGroupMaxLayers];
  int B;
  const int* items;          // [8][cap][2]: (layer, channel-tile pair * 65536 + split) in the order of XCD x's queue
  int cap;
  int count[8];
  unsigned* counter;         // [8], zeroed by the caller before the launch
  int f_es, s_es;            // bytes per element of the F (dY) / S (X) tensors: 2, or 4 = plane pairs, hi plane (WgradParams)
  int max_items;             // items a workgroup takes before it exits and gives its CU back (<= 0: until the queue is empty)
};

// barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the ACKNOWLEDGEMENT of the
// atomics a team-0 wave has just issued -- the very wait this kernel exists to avoid
__device__ __forceinline__ void group_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// a queue word, read as LDS (through a generic pointer a volatile read is a FLAT load: it counts in vmcnt, and the wait the
// compiler puts behind it is a wait for every vector-memory operation outstanding -- the flush included)
__device__ __forceinline__ unsigned lds_word(const unsigned* p) {
  return *(volatile LDS_PTR(unsigned))(LDS_PTR(unsigned))(p);
}

template <typename TA>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_group_kernel(GroupParams gp) {
  constexpr int MODE = 0, TR = 8, NTAPS = 9;
  constexpr int F_ROWS = TR * 16;
  constexpr int S_ROWS = (TR + 2) * 18;
  constexpr int S_ROWS_PAD = (S_ROWS + 7) / 8 * 8;
  constexpr int F_BYTES = F_ROWS * 128;
  constexpr int S_BYTES = S_ROWS_PAD * 128;
  constexpr int BUF_BYTES = F_BYTES + S_BYTES;
  constexpr int NF = F_ROWS / 8 / 4;
  constexpr int NS = (S_ROWS_PAD / 8 + 3) / 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int ws = wave & 1, tg = wave >> 1;
  const int g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  const int sub = lane >> 3, c = lane & 7;
  const int xcd = blockIdx.x & 7;
  // queue words behind the four tile buffers: tq[0..1] team barrier counters, tq[2] next tile of the item, tq[4..7] the
  // tile each team takes next (one iteration ahead), tq[8 + 4 * parity ..]: this workgroup's current / next ITEM:
  // (index in the XCD's queue, layer, channel-tile pair * 65536 + split).
  // The item is fetched by ONE lane of team 1 -- a team-0 wave has its flush outstanding, and any vector-memory load it
  // issued would return behind those atomics (in-order retirement): the whole workgroup would sit through the flush.
  unsigned* tq = reinterpret_cast<unsigned*>(smem + 4 * BUF_BYTES);
  const int n_items = gp.count[xcd];
  auto fetch_item = [&](int par) {
    const unsigned k = __hip_atomic_fetch_add(gp.counter + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned a = 0, b = 0;
    if (k < (unsigned)n_items) {
      const int* it = gp.items + 2 * ((long)xcd * gp.cap + k);
      a = (unsigned)it[0];
      b = (unsigned)it[1];
    }
    tq[8 + 4 * par] = k;
    tq[9 + 4 * par] = a;
    tq[10 + 4 * par] = b;
  };
  if (tid == 256) fetch_item(0);
  __syncthreads();
  int ipar = 0;

  auto run = [&](auto tgc) {
    constexpr int TG = decltype(tgc)::value;
    using G = WgTaps<MODE, TG>;
    f32x4 acc2[G::N][4][2];
    const int RF = (g >> 1) * 16 + 4 * (g & 1) + q;
    const int RSl = (g >> 1) * G::RS + 4 * (g & 1) + q;
    const unsigned lds0 = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem)) + team * 2 * BUF_BYTES;
    const int tt = tid & 255;
    unsigned* tcnt = tq + team;
    for (int taken = 0;; ++taken) {
      const int k = __builtin_amdgcn_readfirstlane((int)lds_word(tq + 8 + 4 * ipar));
      if (k >= n_items) break;
      const int li_ = __builtin_amdgcn_readfirstlane((int)lds_word(tq + 9 + 4 * ipar));
      const int qs = __builtin_amdgcn_readfirstlane((int)lds_word(tq + 10 + 4 * ipar));
      const GroupLayer& p = gp.layer[li_];
      const int qt = qs >> 16, split = qs & 0xFFFF;
      const int cs_tiles = (p.CS + 63) / 64;
      const int cf0 = (qt / cs_tiles) * 64, cs0 = (qt % cs_tiles) * 64;
      const TA* fp = reinterpret_cast<const TA*>(p.f);
      const TA* sp = reinterpret_cast<const TA*>(p.s);
      const int Hs = p.Hf, Ws = p.Wf;
#pragma unroll
      for (int t = 0; t < G::N; ++t)
#pragma unroll
        for (int fh = 0; fh < 4; ++fh)
#pragma unroll
          for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[t][fh][sh][r] = 0.f;
      // per-lane DMA offsets relative to the tile origin (as in wgrad_kernel)
      unsigned f_rel[NF], s_rel[NS];
      auto f_geo = [&](int i, int& ry, int& rx) {
        const int row = 8 * (wave + 4 * i) + sub;
        ry = row >> 4;
        rx = row & 15;
      };
      auto s_geo = [&](int i, int& ry, int& rx) {
        const int row = 8 * (wave + 4 * i) + sub;
        ry = (row * 3641) >> 16;
        rx = row - ry * 18;
      };
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int row = 8 * (wave + 4 * i) + sub;
        const int u = c ^ (swz16(row) >> 4);
        int ry, rx;
        f_geo(i, ry, rx);
        f_rel[i] = (cf0 + u * 8) < p.CF ? (unsigned)(((ry * (long)p.Wf + rx) * p.f_ld + cf0 + u * 8) * gp.f_es) : OOB;
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int kk = wave + 4 * i;
        const int row = 8 * kk + sub;
        const int u = c ^ (swz16(row) >> 4);
        int ry, rx;
        s_geo(i, ry, rx);
        const bool ok = kk < S_ROWS_PAD / 8 && row < S_ROWS && (cs0 + u * 8) < p.CS;
        s_rel[i] = ok ? (unsigned)(((ry * (long)Ws + rx) * p.s_ld + cs0 + u * 8) * gp.s_es) : OOB;
      }
      auto issue_tile = [&](long tile, int buf) {
        const int txi = (int)(tile % p.tiles_x);
        const long ttl = tile / p.tiles_x;
        const int tyi = (int)(ttl % p.tiles_y);
        const long b = ttl / p.tiles_y;
        const int y0 = tyi * TR, x0 = txi * 16;
        unsigned char* base = smem + (team * 2 + buf) * BUF_BYTES;
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(fp) + ((b * p.Hf + y0) * (long)p.Wf + x0) * p.f_ld * gp.f_es), 0,
            0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          int ry, rx;
          f_geo(i, ry, rx);
          const bool ok = (y0 + ry) < p.Hf && (x0 + rx) < p.Wf;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024),
                                                   16, (int)(ok ? f_rel[i] : OOB), 0, 0, 0);
        }
        unsigned char* sbase = base + F_BYTES;
        const int sy0 = y0 - 1, sx0 = x0 - 1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(sp) + ((b * Hs + sy0) * (long)Ws + sx0) * p.s_ld * gp.s_es), 0,
            0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int kk = wave + 4 * i;
          if (kk >= S_ROWS_PAD / 8) continue;          // wave-uniform
          int ry, rx;
          s_geo(i, ry, rx);
          const unsigned y = (unsigned)(sy0 + ry), x = (unsigned)(sx0 + rx);
          const bool ok = y < (unsigned)Hs && x < (unsigned)Ws;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sbase + kk * 1024), 16,
                                                   (int)(ok ? s_rel[i] : OOB), 0, 0, 0);
        }
      };
      auto contract_tile = [&](int cur) {
        const unsigned aF = lds0 + cur * BUF_BYTES, aS = aF + F_BYTES;
        const unsigned fv0 = aF + RF * 128 + 8 * pp + swz16(RF);
        unsigned sv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sv[kk] = aS + RSl * 128 + 8 * pp + ((ws * 64) ^ swz16(kk + RSl));
        WgFrags f;
        wg_rd_a<0, 0>(fv0, f);
        wg_rd_a<0, 1>(fv0, f);
        wg_rd_a<0, 2>(fv0, f);
        wg_rd_a<0, 3>(fv0, f);
        wg_rd_b<MODE, TG, 0, false>(sv, f);
        wg_step<typename PlaneOf<TA>::type, MODE, TG, 0, false>(fv0, sv, f, acc2);
      };
      const long t_begin = (long)split * p.tiles_per_block;
      const long t_end = t_begin + p.tiles_per_block < p.ntiles ? t_begin + p.tiles_per_block : p.ntiles;
      // ---- the two-team tile loop of wgrad_kernel<.., TEAMS = 2> on this item's pixel range --------------------------
      if (tid < 3) tq[tid] = tid == 2 ? 2u : 0u;
      group_barrier_lds();
      unsigned ttarget = 0;
      long cur_tile = t_begin + team;
      bool have = cur_tile < t_end;
      if (have) issue_tile(cur_tile, 0);
      if (tt == 0) tq[4 + 2 * team] = __hip_atomic_fetch_add(tq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // the NEXT item is requested now (two dependent round trips to L2, hidden behind this item's first tile) -- unless
      // this was the workgroup's last one: it then exits and a fresh workgroup (or one of another stream's kernels) gets
      // the CU; a persistent workgroup would hold it until its queue is empty and block every other kernel meanwhile
      const bool last = gp.max_items > 0 && taken + 1 >= gp.max_items;
      if (tid == 256) {
        if (last) tq[8 + 4 * (ipar ^ 1)] = 0x7FFFFFFFu;
        else fetch_item(ipar ^ 1);
      }
      int cur = 0, par = 0;
      while (have) {
        // this wave's share of the tile has landed -- for a team-0 wave this is also where the atomics of the PREVIOUS
        // item are acknowledged (vector-memory operations retire in issue order); team 1 contracts tiles meanwhile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ttarget += 4;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ttarget) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const long nxt = t_begin + __builtin_amdgcn_readfirstlane((int)lds_word(tq + 4 + 2 * team + par));
        par ^= 1;
        if (tt == 0) tq[4 + 2 * team + par] = __hip_atomic_fetch_add(tq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool has_next = nxt < t_end;
        if (has_next) issue_tile(nxt, cur ^ 1);
        contract_tile(cur);
        cur ^= 1;
        cur_tile = nxt;
        have = has_next;
      }
      // team 1's accumulators -> LDS (the tile buffers are dead once every wave has left its loop) -> added to team 0's
      constexpr int N0 = WgTaps<MODE, 0>::N, N1 = WgTaps<MODE, 1>::N;
      static_assert((2 * N0 + 2 * N1) * 8192 <= 4 * BUF_BYTES, "accumulator exchange area");
      unsigned char* xa = smem + (TG == 0 ? ws * N0 : 2 * N0 + ws * N1) * 8192 + lane * 16;
      group_barrier_lds();
      if (team == 1) {
#pragma unroll
        for (int t = 0; t < G::N; ++t)
#pragma unroll
          for (int fh = 0; fh < 4; ++fh)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) *reinterpret_cast<f32x4*>(xa + ((t * 4 + fh) * 2 + sh) * 1024) = acc2[t][fh][sh];
      }
      group_barrier_lds();
      if (team == 0) {
#pragma unroll
        for (int t = 0; t < G::N; ++t)
#pragma unroll
          for (int fh = 0; fh < 4; ++fh)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(xa + ((t * 4 + fh) * 2 + sh) * 1024);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc2[t][fh][sh][r] += o[r];
            }
      }
      // the exchange area has been read: the tile buffers (and the queue words) belong to the next item from here on
      group_barrier_lds();
      ipar ^= 1;
      if (team == 0) {
        // dw[t][cf][cs] += acc, fire and forget: nothing below waits for these (the next tile wait of THIS wave does,
        // in issue order; team 1 keeps the SIMDs busy until then)
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
          const int col = cs0 + ws * 32 + sh * 16 + (lane & 15);
          if (col < p.CS) {
#pragma unroll
            for (int t = 0; t < G::N; ++t)
#pragma unroll
              for (int fh = 0; fh < 4; ++fh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int row = cf0 + fh * 16 + (lane >> 4) * 4 + r;
                  if (row < p.CF) atomicAdd(p.dw + ((long)(G::T0 + t) * p.CF + row) * p.CS + col, acc2[t][fh][sh][r]);
                }
          }
        }
      }
    }
  };
  if (tg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// The same for plane pairs (CRIMAC_PREC_H3P): wgrad_pp_kernel<false>'s tile loop -- one team of 8 waves, waves 4-7 move
// the tiles -- inside the item loop.  All 8 waves flush, so the workgroup does sit through ITS OWN flush (the first tile
// barrier of the next item waits for vmcnt(0)); but that is 147 KB against a memory side that the other 255 workgroups
// are not hitting at the same moment, and the next item's first tile is requested before the flush is issued.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_pp_group_kernel(GroupParams gp) {
  constexpr int TR = 8;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NDMA = 2 * 16 + 2 * 23;
  constexpr int DW = 4;
  constexpr int NI = (NDMA + DW - 1) / DW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = wave & 1, sa = (wave >> 1) & 1, tg = wave >> 2;
  const int sub = lane >> 3, c = lane & 7;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int xcd = blockIdx.x & 7;
  unsigned* tq = reinterpret_cast<unsigned*>(smem + 2 * PP_BUF);      // tq[8 + 4 * parity ..]: (item index, layer, tile pair / split)
  const int n_items = gp.count[xcd];
  auto fetch_item = [&](int par) {
    const unsigned k = __hip_atomic_fetch_add(gp.counter + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned a = 0, b = 0;
    if (k < (unsigned)n_items) {
      const int* it = gp.items + 2 * ((long)xcd * gp.cap + k);
      a = (unsigned)it[0];
      b = (unsigned)it[1];
    }
    tq[8 + 4 * par] = k;
    tq[9 + 4 * par] = a;
    tq[10 + 4 * par] = b;
  };
  if (tid == 256) fetch_item(0);
  __syncthreads();
  int ipar = 0;
  const bool mover = wave >= 4;

  auto run = [&](auto tgc) {
    constexpr int TG = decltype(tgc)::value;
    using G = WgTaps<0, TG>;
    f32x4 acc[G::N][4];
    const int RF = (g >> 1) * 16 + 4 * (g & 1) + q;
    const int RSl = (g >> 1) * G::RS + 4 * (g & 1) + q;
    const unsigned lds0 = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem));
    bool have_prev = false;
    float* prev_dw = nullptr;
    int prev_CF = 0, prev_CS = 0, prev_cf0 = 0, prev_cs0 = 0;
    for (int taken = 0;; ++taken) {
      const int k = __builtin_amdgcn_readfirstlane((int)lds_word(tq + 8 + 4 * ipar));
      const bool valid = k < n_items;
      const int li_ = valid ? __builtin_amdgcn_readfirstlane((int)lds_word(tq + 9 + 4 * ipar)) : 0;
      const int qs = valid ? __builtin_amdgcn_readfirstlane((int)lds_word(tq + 10 + 4 * ipar)) : 0;
      const GroupLayer& p = gp.layer[li_];
      const int qt = qs >> 16, split = qs & 0xFFFF;
      const int cs_tiles = p.CS / 64;
      const int cf0 = (qt / cs_tiles) * 64, cs0 = (qt % cs_tiles) * 64;
      const hp_t* fp = reinterpret_cast<const hp_t*>(p.f);
      const hp_t* sp = reinterpret_cast<const hp_t*>(p.s);
      unsigned rel[NI];
      int lds_at[NI];
      auto geo = [&](int kk_, bool& is_s, int& img, int& row) {
        is_s = kk_ >= 32;
        const int kk = is_s ? kk_ - 32 : kk_;
        img = is_s ? (kk >= 23 ? 1 : 0) : (kk >> 4);
        row = 8 * (is_s ? kk - 23 * img : (kk & 15)) + sub;
      };
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int kk = (wave & (DW - 1)) + DW * i;
        bool is_s; int img, row;
        geo(kk, is_s, img, row);
        const int u = c ^ (swz16(row) >> 4);
        const int su = (((u >> 2) * 2 + (u & 1)) << 1) | ((u >> 1) & 1);
        lds_at[i] = kk < NDMA ? (is_s ? 2 * PP_F_IMG + img * PP_S_IMG + (row - sub) * 128 : img * PP_F_IMG + (row - sub) * 128) : -1;
        if (!is_s) {
          const int ry = row >> 4, rx = row & 15;
          rel[i] = (unsigned)(((ry * (long)p.Wf + rx) * p.f_ld + cf0 + 32 * img) * 4 + su * 16);
        } else {
          const int ry = (row * 3641) >> 16, rx = row - ry * 18;
          rel[i] = row < 180 ? (unsigned)(((ry * (long)p.Wf + rx) * p.s_ld + cs0 + 32 * img) * 4 + su * 16) : OOB;
        }
      }
      auto issue_tile = [&](long tile, int buf) {
        const int txi = (int)(tile % p.tiles_x);
        const long ttl = tile / p.tiles_x;
        const int tyi = (int)(ttl % p.tiles_y);
        const long b = ttl / p.tiles_y;
        const int y0 = tyi * TR, x0 = txi * 16;
        unsigned char* base = smem + buf * PP_BUF;
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<hp_t*>(fp + ((b * p.Hf + y0) * (long)p.Wf + x0) * p.f_ld), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<hp_t*>(sp + ((b * p.Hf + y0 - 1) * (long)p.Wf + x0 - 1) * p.s_ld), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int kk = (wave & (DW - 1)) + DW * i;
          if (kk >= NDMA) continue;                       // wave-uniform
          bool is_s; int img, row;
          geo(kk, is_s, img, row);
          if (!is_s) {
            const bool ok = (y0 + (row >> 4)) < p.Hf && (x0 + (row & 15)) < p.Wf;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                     (int)(ok ? rel[i] : OOB), 0, 0, 0);
          } else {
            const int ry = (row * 3641) >> 16, rx = row - ry * 18;
            const unsigned y = (unsigned)(y0 - 1 + ry), x = (unsigned)(x0 - 1 + rx);
            const bool ok = y < (unsigned)p.Hf && x < (unsigned)p.Wf;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + lds_at[i]), 16,
                                                     (int)(ok ? rel[i] : OOB), 0, 0, 0);
          }
        }
      };
      const long t_begin = (long)split * p.tiles_per_block;
      const long t_end = t_begin + p.tiles_per_block < p.ntiles ? t_begin + p.tiles_per_block : p.ntiles;
      // the next item's first tile is on its way before the finished item's flush is issued
      if (valid && mover && t_begin < t_end) issue_tile(t_begin, 0);
      if (have_prev) {
        // dw[t][cf][cs] += acc of the item just finished: F rows cf0 + 16 fr .. +15, S columns cs0 + 32 sa + 16 ws .. +15
        const int col = prev_cs0 + sa * 32 + ws * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < G::N; ++t)
#pragma unroll
          for (int fr = 0; fr < 4; ++fr)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = prev_cf0 + fr * 16 + (lane >> 4) * 4 + r;
              atomicAdd(prev_dw + ((long)(G::T0 + t) * prev_CF + row) * prev_CS + col, acc[t][fr][r]);
            }
      }
      if (!valid) break;
#pragma unroll
      for (int t = 0; t < G::N; ++t)
#pragma unroll
        for (int fr = 0; fr < 4; ++fr)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][fr][r] = 0.f;
      if (tid == 256) {                              // (a moving wave: its wait is the first tile barrier's anyway)
        if (gp.max_items > 0 && taken + 1 >= gp.max_items) tq[8 + 4 * (ipar ^ 1)] = 0x7FFFFFFFu;      // last item: exit next
        else fetch_item(ipar ^ 1);
      }
      for (long tile = t_begin; tile < t_end; ++tile) {
        const int cur = (int)((tile - t_begin) & 1);
        __syncthreads();           // vmcnt(0) + barrier: the tile has landed for everyone, the other buffer is free
        if (tile + 1 < t_end && mover) issue_tile(tile + 1, cur ^ 1);
        const unsigned aF = lds0 + cur * PP_BUF, aS = aF + 2 * PP_F_IMG + sa * PP_S_IMG;
        const unsigned fv0 = aF + (RF * 128 + 8 * pp + swz16(RF));
        unsigned sv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sv[kk] = aS + RSl * 128 + 8 * pp + ((ws * 64) ^ swz16(kk + RSl));
        PpFrags f;
        pp_rd_a<0, 0>(fv0, f); pp_rd_a<0, 1>(fv0, f); pp_rd_a<0, 2>(fv0, f); pp_rd_a<0, 3>(fv0, f);
        pp_rd_a<0, 4>(fv0, f); pp_rd_a<0, 5>(fv0, f); pp_rd_a<0, 6>(fv0, f); pp_rd_a<0, 7>(fv0, f);
        wg_rd_b<0, TG, 0, false>(sv, f);
        pp_step<TG, 0, 8>(fv0, sv, f, acc);
      }
      group_barrier_lds();         // every wave has read the item's last tile: both buffers belong to the next item
      prev_dw = p.dw; prev_CF = p.CF; prev_CS = p.CS; prev_cf0 = cf0; prev_cs0 = cs0;
      have_prev = true;
      ipar ^= 1;
    }
  };
  if (tg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// Pixel-range split of a shape: fills tiles_y/x, ntiles, tiles_per_block, nsplits (same for every precision).
// teams: 2 for the 8-wave kernel (one workgroup per CU), 1 for the 4-wave kernel (two per CU).
void plan_splits(WgradParams& p, int mode, int target_blocks, int teams) {
  const int TR = mode == 0 ? 8 : 4;
  p.tiles_y = cdiv(p.Hf, TR);
  p.tiles_x = cdiv(p.Wf, 16);
  p.ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  const int ch_tiles = cdiv(p.CF, 64) * cdiv(p.CS, 64);
  const bool autosplit = target_blocks <= 0;
  // one resident round: 2 workgroups per CU x 256 CUs.  More splits only add partial-sum volume (each workgroup
  // ends with 147 KB of fp32 output) and a second, partially filled round: 512 measured 2 % faster than 1024.
  static const int auto_blocks = getenv("CRIMAC_WGRAD_BLOCKS") ? atoi(getenv("CRIMAC_WGRAD_BLOCKS")) : 512;
  if (autosplit) target_blocks = auto_blocks / teams;
  const int round = 512 / teams;               // workgroups of one resident round
  int splits = target_blocks / ch_tiles;
  if (autosplit) {
    // every split adds one pass over dW (fp32 atomics at ~1.3 TB/s chip-wide, or a partial slab written and read
    // back): keep at least ~4096 contraction pixels per split so that pass stays below the MFMA time
    long max_splits = ((long)p.B * p.Hf * p.Wf) / 4096;
    if (max_splits < 1) max_splits = 1;
    // ... unless that leaves fewer than two workgroups per CU: then parallelism is worth more
    while (max_splits * ch_tiles < round && max_splits * 2 <= ((long)p.B * p.Hf * p.Wf) / 1024) max_splits *= 2;
    if (splits > max_splits) splits = (int)max_splits;
  }
  if (splits < 1) splits = 1;
  if (splits > p.ntiles) splits = (int)p.ntiles;
  p.tiles_per_block = cdiv(p.ntiles, splits);
  p.nsplits = cdiv(p.ntiles, p.tiles_per_block);       // every split owns at least one tile
}

template <typename TA, int NPL, int MODE, bool NARROW = false, int TEAMS = 1>
int launch(WgradParams p, int target_blocks, hipStream_t st) {
  constexpr int TR = MODE == 0 ? 8 : 4;
  constexpr int F_ROWS = TR * 16;
  constexpr int S_ROWS = ((MODE == 0 ? (TR + 2) * 18 : (2 * TR) * 32) + 7) / 8 * 8;   // padded to 8 rows
  plan_splits(p, MODE, target_blocks, TEAMS);
  const int ch_tiles = cdiv(p.CF, 64) * cdiv(p.CS, 64);
  const int splits = p.nsplits;
  // bf16: two buffers per team (double-buffered direct-to-LDS tiles, + the tile queue); fp32: one buffer of NPL planes
  const size_t lds = (size_t)(F_ROWS + S_ROWS) * 128 * (NPL == 1 ? 2 * TEAMS : NPL) + (TEAMS == 2 ? 64 : 0);
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TA, NPL, MODE, NARROW, TEAMS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((wgrad_kernel<TA, NPL, MODE, NARROW, TEAMS>), dim3(ch_tiles * splits), dim3(256 * TEAMS), lds, st, p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

int launch_pp(WgradParams p, int target_blocks, hipStream_t st) {
  plan_splits(p, 0, target_blocks, 2);           // (one 8-wave workgroup per CU, like the two-team kernel)
  const bool narrow = p.CS == 16;
  const int ch_tiles = (p.CF / 64) * (narrow ? 1 : p.CS / 64);
  const size_t lds = 2 * (size_t)PP_BUF;
  static unsigned long long attr_devs = 0;
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  if (narrow) hipLaunchKernelGGL(wgrad_pp_kernel<true>, dim3(ch_tiles * p.nsplits), dim3(512), lds, st, p);
  else hipLaunchKernelGGL(wgrad_pp_kernel<false>, dim3(ch_tiles * p.nsplits), dim3(512), lds, st, p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// plane pairs, transposed convolution: the 128 x 64 kernel above (CRIMAC_WGRAD_UP_PP=0: the register-staged kernel)
bool up_pp_ok(int prec, int mode, int CF, int CS, long f_ld, long s_ld, int Wf, long partial_stride) {
  static const int off = getenv("CRIMAC_WGRAD_UP_PP") ? atoi(getenv("CRIMAC_WGRAD_UP_PP")) == 0 : 0;
  return !off && prec == CRIMAC_PREC_H3P && mode == 1 && partial_stride == 0 && CF % 128 == 0 && CS % 64 == 0 &&
         (2L * Wf + 16) * f_ld * 4 < (1L << 31) && (8L * Wf + 32) * s_ld * 4 < (1L << 31);
}
int launch_up_pp(WgradParams p, int target_blocks, hipStream_t st) {
  p.tiles_y = cdiv(p.Hf, 2);
  p.tiles_x = cdiv(p.Wf, 16);
  p.ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  const int ch_tiles = (p.CF / 128) * (p.CS / 64);
  const int target = target_blocks > 0 ? target_blocks : 256;       // one workgroup of 8 waves per CU
  long splits = target / ch_tiles;
  if (splits < 1) splits = 1;
  if (splits > p.ntiles) splits = p.ntiles;
  p.tiles_per_block = cdiv(p.ntiles, splits);
  p.nsplits = cdiv(p.ntiles, p.tiles_per_block);
  static unsigned long long attr_devs = 0;
  if (crimac_first_use_on_device(&attr_devs))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_up_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  hipLaunchKernelGGL(wgrad_up_pp_kernel, dim3(ch_tiles * p.nsplits), dim3(512), 2 * (size_t)UP_BUF, st, p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

// 16-bit storage, conv3x3: the two-team kernel (CRIMAC_WGRAD_TEAMS=1 selects the 4-wave kernel for A/B runs)
// plane pairs, conv3x3, whole 64-channel tiles both ways, 32-bit DMA offsets inside a tile: the 8-wave kernel above
bool pp_ok(int prec, int mode, int CF, int CS, long f_ld, long s_ld, int Wf) {
  static const int off = getenv("CRIMAC_WGRAD_PP") ? atoi(getenv("CRIMAC_WGRAD_PP")) == 0 : 0;
  return !off && prec == CRIMAC_PREC_H3P && mode == 0 && CF % 64 == 0 && (CS % 64 == 0 || CS == 16) &&
         (10L * Wf + 18) * (f_ld > s_ld ? f_ld : s_ld) * 4 < (1L << 31);
}
int teams_of(int prec, int mode) {
  static const int forced = getenv("CRIMAC_WGRAD_TEAMS") ? atoi(getenv("CRIMAC_WGRAD_TEAMS")) : 0;
  if (forced == 1) return 1;
  return (prec == CRIMAC_PREC_BF16 || prec == CRIMAC_PREC_FP16) && mode == 0 ? 2 : 1;
}

int wgrad_run(int prec, int mode, const void* f, long f_ld, int CF, const void* s, long s_ld, int CS, int B, int Hf,
              int Wf, float* dw, long partial_stride, int target_blocks, void* stream) {
  const bool mix = prec == CRIMAC_PREC_H3F_BWD;
  if (mix) {
    // fp16 contraction; the ACTIVATION operand (S for the convolution, F for the transposed convolution) is an fp16
    // plane-pair tensor read through its hi plane, the GRADIENT operand plain fp16
    CRIMAC_REQUIRE(partial_stride == 0, "wgrad (H3F_BWD): the slab form is not available");
    prec = CRIMAC_PREC_FP16;
  }
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, "wgrad: bad precision %d", prec);
  CRIMAC_REQUIRE(prec != CRIMAC_PREC_F32H3, "wgrad: F32H3 is a forward-operand mode (fp16 planes have no range for "
                 "unscaled gradients); call the backward kernels with CRIMAC_PREC_F32X3, or use CRIMAC_PREC_H3P");
  CRIMAC_REQUIRE(mode == 0 || mode == 1, "wgrad: bad mode %d", mode);
  CRIMAC_REQUIRE(CF > 0 && CF % 8 == 0 && CS > 0 && CS % 8 == 0, "wgrad: channels must be multiples of 8");
  CRIMAC_REQUIRE(f_ld >= CF && s_ld >= CS && f_ld % 8 == 0 && s_ld % 8 == 0, "wgrad: bad pixel strides");
  CRIMAC_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && f && s && dw, "wgrad: bad arguments");
  CRIMAC_REQUIRE(partial_stride == 0 || partial_stride >= (long)(mode == 0 ? 9 : 4) * CF * CS,
                 "wgrad: partial slab stride %ld smaller than one dW (%d taps x %d x %d)", partial_stride,
                 mode == 0 ? 9 : 4, CF, CS);
  WgradParams p;
  p.f = f; p.f_ld = f_ld; p.CF = CF; p.s = s; p.s_ld = s_ld; p.CS = CS;
  p.B = B; p.Hf = Hf; p.Wf = Wf; p.dw = dw; p.partial_stride = partial_stride;
  if (mix) {
    if (mode == 0) p.s_es = 4; else p.f_es = 4;
    const long ldmax = f_ld > s_ld ? f_ld : s_ld;
    CRIMAC_REQUIRE((20L * Wf + 64) * ldmax * 4 < (1L << 31), "wgrad (H3F_BWD): a tile's DMA offsets exceed 32 bits");
  }
  hipStream_t st = (hipStream_t)stream;
  const bool two = teams_of(prec, mode) == 2;
  if (prec == CRIMAC_PREC_BF16) {
    if (mode == 0 && CS <= 16)                                                              // first layer
      return two ? launch<bf16_t, 1, 0, true, 2>(p, target_blocks, st) : launch<bf16_t, 1, 0, true>(p, target_blocks, st);
    if (mode == 0) return two ? launch<bf16_t, 1, 0, false, 2>(p, target_blocks, st) : launch<bf16_t, 1, 0>(p, target_blocks, st);
    return launch<bf16_t, 1, 1>(p, target_blocks, st);
  }
  if (prec == CRIMAC_PREC_FP16) {
    if (mode == 0 && CS <= 16)
      return two ? launch<half_t, 1, 0, true, 2>(p, target_blocks, st) : launch<half_t, 1, 0, true>(p, target_blocks, st);
    if (mode == 0) return two ? launch<half_t, 1, 0, false, 2>(p, target_blocks, st) : launch<half_t, 1, 0>(p, target_blocks, st);
    return launch<half_t, 1, 1>(p, target_blocks, st);
  }
  if (prec == CRIMAC_PREC_F32X3)
    return mode == 0 ? launch<float, 2, 0>(p, target_blocks, st) : launch<float, 2, 1>(p, target_blocks, st);
  if (prec == CRIMAC_PREC_H3P) {    // both operands are fp16 plane pairs (activation x loss-scaled output gradient)
    if (pp_ok(prec, mode, CF, CS, f_ld, s_ld, Wf)) return launch_pp(p, target_blocks, st);
    if (up_pp_ok(prec, mode, CF, CS, f_ld, s_ld, Wf, p.partial_stride)) return launch_up_pp(p, target_blocks, st);
    return mode == 0 ? launch<hp_t, 2, 0>(p, target_blocks, st) : launch<hp_t, 2, 1>(p, target_blocks, st);
  }
  return mode == 0 ? launch<float, 3, 0>(p, target_blocks, st) : launch<float, 3, 1>(p, target_blocks, st);
}

}  // namespace

// ---- grouped launch: plan (host) and launch ------------------------------------------------------------------------------
static int group_check(int prec, const crimac_wgrad_group_layer* L, int n, int B) {
  CRIMAC_REQUIRE(prec == CRIMAC_PREC_BF16 || prec == CRIMAC_PREC_FP16 || prec == CRIMAC_PREC_H3P || prec == CRIMAC_PREC_H3F_BWD,
                 "wgrad_group: 16-bit storage precisions, plane pairs and H3F_BWD only (prec=%d)", prec);
  const bool hp = prec == CRIMAC_PREC_H3P;
  const bool mix = prec == CRIMAC_PREC_H3F_BWD;       // (S = the activation: plane pairs addressed in 4-byte elements)
  CRIMAC_REQUIRE(L && n >= 1 && n <= CRIMAC_WGRAD_GROUP_MAX_LAYERS && B > 0, "wgrad_group: 1..%d layers", CRIMAC_WGRAD_GROUP_MAX_LAYERS);
  for (int i = 0; i < n; ++i) {
    CRIMAC_REQUIRE(L[i].CF > 0 && L[i].CF % 8 == 0 && L[i].CS >= 64 && L[i].CS % 8 == 0, "wgrad_group: layer %d: CF=%d CS=%d "
                   "(CS >= 64: the first layer keeps its own launch)", i, L[i].CF, L[i].CS);
    CRIMAC_REQUIRE(L[i].f_ld >= L[i].CF && L[i].s_ld >= L[i].CS && L[i].f_ld % 8 == 0 && L[i].s_ld % 8 == 0 && L[i].Hf > 0 && L[i].Wf > 0,
                   "wgrad_group: layer %d: bad strides / sizes", i);
    CRIMAC_REQUIRE((10L * L[i].Wf + 18) * (L[i].f_ld > L[i].s_ld ? L[i].f_ld : L[i].s_ld) * ((hp || mix) ? 4 : 2) < (1L << 31),
                   "wgrad_group: layer %d: a tile's DMA offsets exceed 32 bits", i);
    CRIMAC_REQUIRE(!hp || (L[i].CF % 64 == 0 && L[i].CS % 64 == 0), "wgrad_group: layer %d: plane pairs need whole 64-channel "
                   "tiles (CF=%d CS=%d)", i, L[i].CF, L[i].CS);
  }
  return CRIMAC_OK;
}

extern "C" int crimac_wgrad_group_layer_size(void) { return (int)sizeof(crimac_wgrad_group_layer); }

extern "C" int crimac_wgrad_group_plan(int prec, crimac_wgrad_group_layer* layers, int n_layers, int B, int items_per_layer,
                                       int* items, int cap, int* counts) {
  if (int rc = group_check(prec, layers, n_layers, B)) return rc;
  CRIMAC_REQUIRE(counts, "wgrad_group_plan: counts is NULL");
  static const int env_items = getenv("CRIMAC_WGRAD_GROUP_ITEMS") ? atoi(getenv("CRIMAC_WGRAD_GROUP_ITEMS")) : 0;
  // 128 items per layer: half the splits (and half the atomic volume) of one launch per layer, whose 256 splits were
  // there to fill the chip; measured equal to 256 within noise (bf16 step 11.73 vs 11.78 ms), 384 slower (11.91)
  const int target = items_per_layer > 0 ? items_per_layer : (env_items > 0 ? env_items : 128);
  int order[CRIMAC_WGRAD_GROUP_MAX_LAYERS];
  for (int i = 0; i < n_layers; ++i) {
    crimac_wgrad_group_layer& l = layers[i];
    l.tiles_y = cdiv(l.Hf, 8);
    l.tiles_x = cdiv(l.Wf, 16);
    l.ntiles = (long)B * l.tiles_y * l.tiles_x;
    const int ch_tiles = cdiv(l.CF, 64) * cdiv(l.CS, 64);
    long splits = target / ch_tiles;
    // every split costs one atomic pass over its 64 x 64 x 9 tile: keep ~4096 contraction pixels per split unless that
    // leaves the layer with too few items to spread over the chip
    const long pix = (long)B * l.Hf * l.Wf;
    long max_splits = pix / 4096 > 1 ? pix / 4096 : 1;
    while (max_splits * ch_tiles < target / 2 && max_splits * 2 <= pix / 1024) max_splits *= 2;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > l.ntiles) splits = l.ntiles;
    if (splits >= 8) splits = splits / 8 * 8;               // whole rounds of the 8 XCDs
    l.tiles_per_block = cdiv(l.ntiles, splits);
    l.nsplits = cdiv(l.ntiles, l.tiles_per_block);
    CRIMAC_REQUIRE(ch_tiles < 65536 && l.nsplits < 65536, "wgrad_group_plan: layer %d does not fit the item encoding", i);
    order[i] = i;
  }
  // longest items first (a queue of unequal items balances best that way); equal shapes keep their order
  for (int a = 1; a < n_layers; ++a)
    for (int b = a; b > 0 && layers[order[b]].tiles_per_block > layers[order[b - 1]].tiles_per_block; --b) {
      const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t;
    }
  int n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long rr = 0;
  for (int oi = 0; oi < n_layers; ++oi) {
    const int li = order[oi];
    const crimac_wgrad_group_layer& l = layers[li];
    const int ch_tiles = cdiv(l.CF, 64) * cdiv(l.CS, 64);
    auto put = [&](int x, int qt, int split) {
      if (items && n[x] < cap) {
        items[2 * ((long)x * cap + n[x])] = li;
        items[2 * ((long)x * cap + n[x]) + 1] = qt * 65536 + split;
      }
      ++n[x];
    };
    if (l.nsplits % 8 == 0) {
      // split-major as in wgrad_kernel: XCD x owns the pixel ranges x, x + 8, ... and runs ALL channel-tile pairs on
      // them back to back, so a staged F / S tile is fetched into that XCD's L2 once
      for (int x = 0; x < 8; ++x)
        for (int sp = x; sp < l.nsplits; sp += 8)
          for (int qt = 0; qt < ch_tiles; ++qt) put(x, qt, sp);
    } else {
      for (int sp = 0; sp < l.nsplits; ++sp)
        for (int qt = 0; qt < ch_tiles; ++qt) put((int)(rr++ % 8), qt, sp);
    }
  }
  int need = 0;
  for (int x = 0; x < 8; ++x) { counts[x] = n[x]; if (n[x] > need) need = n[x]; }
  if (items) CRIMAC_REQUIRE(need <= cap, "wgrad_group_plan: %d items per queue, room for %d", need, cap);
  return need;              // (>= 0: the queue capacity this plan needs; sizing call: items == NULL)
}

extern "C" int crimac_wgrad_group(int prec, const crimac_wgrad_group_layer* layers, int n_layers, int B, const int* items,
                                  int cap, const int* counts, unsigned int* counters, void* stream) {
  if (int rc = group_check(prec, layers, n_layers, B)) return rc;
  CRIMAC_REQUIRE(items && counts && counters && cap > 0, "wgrad_group: bad queue arguments");
  GroupParams gp;
  for (int i = 0; i < n_layers; ++i) {
    gp.layer[i] = layers[i];
    CRIMAC_REQUIRE(layers[i].f && layers[i].s && layers[i].dw, "wgrad_group: layer %d: NULL tensor", i);
    CRIMAC_REQUIRE(layers[i].tiles_per_block > 0 && layers[i].nsplits > 0 && layers[i].ntiles == (long)B * layers[i].tiles_y * layers[i].tiles_x &&
                       layers[i].tiles_y == cdiv(layers[i].Hf, 8) && layers[i].tiles_x == cdiv(layers[i].Wf, 16),
                   "wgrad_group: layer %d was not planned for this geometry (crimac_wgrad_group_plan)", i);
  }
  gp.B = B; gp.items = items; gp.cap = cap; gp.counter = counters;
  gp.f_es = 2; gp.s_es = prec == CRIMAC_PREC_H3F_BWD ? 4 : 2;
  long total = 0;
  for (int x = 0; x < 8; ++x) {
    CRIMAC_REQUIRE(counts[x] >= 0 && counts[x] <= cap, "wgrad_group: queue %d holds %d items, capacity %d", x, counts[x], cap);
    gp.count[x] = counts[x];
    total += counts[x];
  }
  if (total == 0) return CRIMAC_OK;
  constexpr size_t lds = 4 * (size_t)((8 * 16 + ((8 + 2) * 18 + 7) / 8 * 8) * 128) + 64;
  hipStream_t st = (hipStream_t)stream;
  int ncu = crimac_cu_count();
  ncu = ncu / 8 * 8;                       // (equal shares of the 8 XCD queues)
  if (ncu < 8) ncu = 8;
  // Workgroups take `wg_items` items each and exit: the launch is then as many workgroups as that takes (the hardware
  // keeps one per CU resident and starts the next as one retires), and kernels of OTHER streams get CUs in between -- a
  // fully persistent grid (CRIMAC_WGRAD_GROUP_WGITEMS=0) holds every CU until its queue is empty, which blocks the input-
  // gradient chain on the caller's stream for the whole launch.  Measured, bf16 step / serialized sum of the weight-
  // gradient launches: persistent 11.79 ms / 2.90 ms, 3 items 11.70 / 3.65 (the last round of workgroups is ragged --
  // in the step the other stream's kernels fill it, alone it is idle CUs), 2 items 11.80 / 3.13, 1 item 11.77 / 2.93;
  // one launch per layer 11.93 / 3.19.  One item per workgroup: nothing left of "persistent" but the shared queue -- the
  // flush of a finished workgroup still drains under its neighbours' MFMAs, which is what the grouping is for.
  static const int wg_items = getenv("CRIMAC_WGRAD_GROUP_WGITEMS") ? atoi(getenv("CRIMAC_WGRAD_GROUP_WGITEMS")) : 1;
  gp.max_items = wg_items;
  if (wg_items > 0) {
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) { const int n = cdiv(counts[x], wg_items); if (n > per_xcd) per_xcd = n; }
    ncu = 8 * per_xcd;
  }
  if (prec == CRIMAC_PREC_H3P) {
    static unsigned long long attr_devs = 0;
    if (crimac_first_use_on_device(&attr_devs))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pp_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(wgrad_pp_group_kernel, dim3(ncu), dim3(512), 2 * (size_t)PP_BUF + 64, st, gp);
  } else if (prec == CRIMAC_PREC_BF16) {
    static unsigned long long attr_devs = 0;
    if (crimac_first_use_on_device(&attr_devs))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(wgrad_group_kernel<bf16_t>, dim3(ncu), dim3(512), lds, st, gp);
  } else {                                 // fp16, and H3F_BWD (fp16 contraction, S read from plane pairs)
    static unsigned long long attr_devs = 0;
    if (crimac_first_use_on_device(&attr_devs))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<half_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(wgrad_group_kernel<half_t>, dim3(ncu), dim3(512), lds, st, gp);
  }
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

extern "C" int crimac_wgrad(int prec, int mode, const void* f, long f_ld, int CF, const void* s,
                            long s_ld, int CS, int B, int Hf, int Wf, float* dw, int target_blocks,
                            void* stream) {
  return wgrad_run(prec, mode, f, f_ld, CF, s, s_ld, CS, B, Hf, Wf, dw, 0, target_blocks, stream);
}

extern "C" int crimac_wgrad_splits(int prec, int mode, int CF, int CS, int B, int Hf, int Wf, int target_blocks) {
  if (prec < CRIMAC_PREC_BF16 || prec > CRIMAC_PREC_MAX || !(mode == 0 || mode == 1) || CF <= 0 || CS <= 0 || B <= 0 ||
      Hf <= 0 || Wf <= 0)
    return CRIMAC_ERR_INVALID;
  WgradParams p;
  p.CF = CF; p.CS = CS; p.B = B; p.Hf = Hf; p.Wf = Wf;
  plan_splits(p, mode, target_blocks, pp_ok(prec, mode, CF, CS, CF, CS, Wf) ? 2 : teams_of(prec, mode));
  return p.nsplits;
}

extern "C" int crimac_wgrad_partials(int prec, int mode, const void* f, long f_ld, int CF, const void* s,
                                     long s_ld, int CS, int B, int Hf, int Wf, float* partials, long slab_stride,
                                     int target_blocks, void* stream) {
  CRIMAC_REQUIRE(slab_stride > 0, "wgrad_partials: slab_stride must be positive");
  return wgrad_run(prec, mode, f, f_ld, CF, s, s_ld, CS, B, Hf, Wf, partials, slab_stride, target_blocks, stream);
}
