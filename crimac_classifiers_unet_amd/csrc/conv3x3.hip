// 3x3 convolution (stride 1, pad 1) as an im2col-free implicit GEMM on MFMA, CDNA4 (gfx950).
//
// Reference op: nn.Conv2d(k=3, padding=1, bias=True) forward (unet.py:35-44; call sites :77, :80,
// :114-119) and its input gradient (aten::convolution_backward data half, pipeline.py:177 -- the
// same kernel run on dY with flipped/transposed weights).
//
// Structure (one 256-thread workgroup = 4 waves as 2x2):
//   * output tile = 8x16 spatial patch (128 pixels) x BN output channels; each wave 64 x BN/2 via
//     v_mfma_f32_32x32x16_bf16;
//   * the input HALO tile (10x18 pixels x BK channels) is staged ONCE per input-channel chunk into
//     LDS and read 9 times at shifted rows (one per tap): ~6.4x less global->LDS traffic for the
//     activation operand than gathering per tap, no im2col buffer anywhere;
//   * the weight tile [BN x BK] of step (chunk, tap) streams through a 2-slot LDS ring, fetched to
//     registers TWO steps ahead of its use, so HBM/L2 latency is covered by two steps of MFMAs; the
//     next chunk's halo is fetched at tap 0 and committed to the other LDS buffer at tap 7;
//   * one barrier per step; LDS images are XOR-swizzled so the 16-byte fragment reads are
//     bank-conflict free (2 of 16 lanes 2-way on the shifted halo rows);
//   * epilogue: bias (+ReLU) in registers, tile staged through LDS and written with coalesced
//     16-byte stores; optionally the per-channel sum / sum of squares of the stored tile
//     (BatchNorm batch statistics, unet.py:78,81,121-122) are reduced in LDS and added to fp64
//     accumulators -- the BN statistics pass over the conv output is fused away.
//
// Precision: TA = bf16_t -> one MFMA per product; TA = float -> fp32 activations split into bf16
// hi+lo while staging, weights pre-split, 3 MFMAs per product (see igemm.hip).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

namespace {

struct ConvParams {
  const void* in;
  long in_ld;
  int B, H, W;
  int Cin, N;
  const unsigned short* w_hi;
  const unsigned short* w_lo;
  int w_il;                    // weight planes interleaved in w_hi (CRIMAC_PLANES_INTERLEAVED); else w_hi | w_lo planes
  EpiParams epi;
  int tiles_y, tiles_x;
};

constexpr int TC = 16, HP = TC + 2;       // tile columns, halo pitch

// XOR swizzle of the 16-byte units inside a row (RB bytes per row, RPL rows per 256-byte bank line).
// Weight tile: key from the row index (a 16-lane read group covers 16 consecutive-ish rows).
// Halo tile: key from the halo COLUMN hx (row = hy*18 + hx): a ds_read_b128 lane group of the A
// fragment holds pixels {0-3,12-15} of one image row and {4-11} of the next, whose linear halo rows
// collide mod 16 but whose columns do not -- conflict-free for every tap shift (HP even => the
// bank-line half of a row is hx & 1 for BK = 64).
template <int BK> struct Sw {
  static constexpr int RB = BK * 2;
  static constexpr int UPR = BK / 8;
  static constexpr int RPL = 256 / RB;
  __device__ static __forceinline__ int off(int row, int u) {
    return row * RB + ((u ^ ((row / RPL) % UPR)) << 4);
  }
  __device__ static __forceinline__ int off_halo(int row, int hx, int u) {
    return row * RB + ((u ^ ((hx / RPL) % UPR)) << 4);
  }
};

// TR = tile rows: 8 (128-pixel tile, wave 64 x BN/2, 2 workgroups per CU) or 16 (256-pixel tile,
// wave 128 x BN/2: 25 % fewer LDS fragment bytes per MFMA, 1 workgroup per CU).
// TA = hp_t: the input is already split (fp16 plane pairs, common.h): the staging copies the two 16-byte halves of
// an 8-channel group into the two plane images instead of splitting fp32 values.  TO: storage type of the output.
template <typename TA, int NPL, int BN, int BK, int TR, typename P16 = typename PlaneOf<TA>::type, typename TO = TA>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvParams p) {
  constexpr int HALO_ROWS = (TR + 2) * HP;
  constexpr int BM = TR * TC;
  constexpr int MT = BM / 64;                            // 32-row MFMA tiles per wave
  constexpr bool X3 = sizeof(TA) == 4;                  // fp32 activations, split into NPL bf16 planes
  constexpr bool PRE = __is_same(TA, hp_t);             // ... or plane pairs split by the producer
  static_assert(X3 ? (NPL == 2 || NPL == 3) : NPL == 1, "bf16 -> 1 plane, fp32 -> 2 or 3 planes");
  static_assert(!PRE || NPL == 2, "plane pairs are two planes");
  constexpr int UPR = BK / 8;
  constexpr int NU_H = (HALO_ROWS * UPR + 255) / 256;   // halo units per thread
  constexpr int NU_B = (BN * UPR + 255) / 256;
  constexpr bool B_GUARD = (BN * UPR) % 256 != 0;
  constexpr int NT = BN / 64;
  constexpr int KS = BK / 16;
  constexpr int A_BYTES = HALO_ROWS * BK * 2;           // one plane, one buffer
  constexpr int B_BYTES = BN * BK * 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [buffer][plane] images
  auto sA = [&](int buf, int pl) { return smem + (buf * NPL + pl) * A_BYTES; };
  auto sB = [&](int buf, int pl) { return smem + 2 * NPL * A_BYTES + (buf * NPL + pl) * B_BYTES; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  const int tilesN = p.N / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int tile_n = bid % tilesN;
  int tile_m = bid / tilesN;
  const int txi = tile_m % p.tiles_x;
  tile_m /= p.tiles_x;
  const int tyi = tile_m % p.tiles_y;
  const int b = tile_m / p.tiles_y;
  const int y0 = tyi * TR, x0 = txi * TC, n0 = tile_n * BN;

  const TA* inp = reinterpret_cast<const TA*>(p.in);
  const long w_plane = 9L * p.N * p.Cin;      // elements per weight plane

  // ---- halo staging coordinates (independent of the channel chunk) -----------------------------
  long h_off[NU_H];
  int h_lds[NU_H];
  bool h_ok[NU_H];
#pragma unroll
  for (int i = 0; i < NU_H; ++i) {
    const int q = tid + 256 * i;
    const int row = q / UPR, u = q % UPR;
    const int y = y0 + row / HP - 1, x = x0 + row % HP - 1;
    h_ok[i] = q < HALO_ROWS * UPR && y >= 0 && y < p.H && x >= 0 && x < p.W;
    h_off[i] = (((long)b * p.H + y) * p.W + x) * p.in_ld + u * 8;
    h_lds[i] = q < HALO_ROWS * UPR ? Sw<BK>::off_halo(row, row % HP, u) : -1;
  }

  u32x4 rh[NU_H][X3 ? 2 : 1];   // halo prefetch registers (fp32: 8 values = 2 x 16 B)
  u32x4 rb[2][NU_B][NPL];       // weight prefetch registers (NPL planes), two steps deep

  auto load_halo = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NU_H; ++i) {
      if (h_ok[i]) {
        const TA* src = inp + h_off[i] + kc * BK;
        rh[i][0] = *reinterpret_cast<const u32x4*>(src);
        if constexpr (X3) rh[i][1] = *reinterpret_cast<const u32x4*>(src + 4);
      } else {
        rh[i][0] = u32x4{0, 0, 0, 0};
        if constexpr (X3) rh[i][1] = u32x4{0, 0, 0, 0};
      }
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NU_H; ++i) {
      if (h_lds[i] < 0) continue;
      if constexpr (PRE) {
        *reinterpret_cast<u32x4*>(sA(buf, 0) + h_lds[i]) = rh[i][0];
        *reinterpret_cast<u32x4*>(sA(buf, 1) + h_lds[i]) = rh[i][1];
      } else if constexpr (X3) {
        u32x4 pl[NPL];
        split8<NPL, P16>(rh[i][0], rh[i][1], pl);
#pragma unroll
        for (int k = 0; k < NPL; ++k) *reinterpret_cast<u32x4*>(sA(buf, k) + h_lds[i]) = pl[k];
      } else {
        *reinterpret_cast<u32x4*>(sA(buf, 0) + h_lds[i]) = rh[i][0];
      }
    }
  };
  auto load_b = [&](auto set, int kc, int t) {
#pragma unroll
    for (int i = 0; i < NU_B; ++i) {
      const int q = tid + 256 * i;
      if (B_GUARD && q >= BN * UPR) continue;
      const int row = q / UPR, u = q % UPR;
      if (NPL == 2 && p.w_il) {                  // interleaved plane pairs: [CB hi | CB lo] per block of CB channels
        const long off = ((long)t * p.N + n0 + row) * 2 * p.Cin + il_pos(kc * BK + u * 8, p.Cin);
        rb[set][i][0] = *reinterpret_cast<const u32x4*>(p.w_hi + off);
        rb[set][i][NPL - 1] = *reinterpret_cast<const u32x4*>(p.w_hi + off + il_cb(p.Cin));
        continue;
      }
      const long off = ((long)t * p.N + n0 + row) * p.Cin + kc * BK + u * 8;
      rb[set][i][0] = *reinterpret_cast<const u32x4*>(p.w_hi + off);
#pragma unroll
      for (int k = 1; k < NPL; ++k)
        rb[set][i][k] = *reinterpret_cast<const u32x4*>(p.w_lo + (long)(k - 1) * w_plane + off);
    }
  };
  auto store_b = [&](auto set, int buf) {
#pragma unroll
    for (int i = 0; i < NU_B; ++i) {
      const int q = tid + 256 * i;
      if (B_GUARD && q >= BN * UPR) continue;
      const int o = Sw<BK>::off(q / UPR, q % UPR);
#pragma unroll
      for (int k = 0; k < NPL; ++k) *reinterpret_cast<u32x4*>(sB(buf, k) + o) = rb[set][i][k];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's two A rows (pixels) of the tile: m = wr*64 + i*32 + fr -> (py, px)
  int a_row0[MT], a_hx0[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = wr * (BM / 2) + i * 32 + fr;
    a_row0[i] = (m >> 4) * HP + (m & 15);      // halo row of tap (0,0)
    a_hx0[i] = m & 15;                          // its halo column
  }

  auto compute = [&](int abuf, int bbuf, int t) {
    const int kx = t % 3;
    const int shift = (t / 3) * HP + kx;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[MT][NPL], bfr[NT][NPL];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int o = Sw<BK>::off_halo(a_row0[i] + shift, a_hx0[i] + kx, 2 * ks + fh);
#pragma unroll
        for (int k = 0; k < NPL; ++k) af[i][k] = *reinterpret_cast<const bf16x8*>(sA(abuf, k) + o);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int o = Sw<BK>::off(wc * (BN / 2) + j * 32 + fr, 2 * ks + fh);
#pragma unroll
        for (int k = 0; k < NPL; ++k) bfr[j][k] = *reinterpret_cast<const bf16x8*>(sB(bbuf, k) + o);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mfma_planes<NPL, P16>(af[i], bfr[j], acc[i][j]);
    }
  };

  const int kchunks = p.Cin / BK;
  const int nsteps = kchunks * 9;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue: halo(0), B(0) in LDS; B(1) in flight -------------------------------------------
  load_halo(0);
  load_b(I0{}, 0, 0);
  store_halo(0);
  store_b(I0{}, 0);
  if (nsteps > 1) load_b(I1{}, 0, 1);      // 9 taps per chunk: step 1 is (kc 0, tap 1)
  __syncthreads();

  int kc = 0, t = 0;       // coordinates of the current step s
  // One step; PAR = s % 2 selects the register set / LDS slot statically.
  auto step = [&](auto PAR, int s) {
    constexpr int par = decltype(PAR)::value;
    using SetCur = std::integral_constant<int, par>;        // set that will hold B(s+2)
    using SetNext = std::integral_constant<int, 1 - par>;   // set holding B(s+1)
    // coordinates of steps s+2
#ifndef CRIMAC_EXP_NOLOAD
    if (s + 2 < nsteps) {
      int t2 = t + 2, kc2 = kc;
      if (t2 >= 9) { t2 -= 9; kc2 += 1; }
      load_b(SetCur{}, kc2, t2);
    }
#endif
    const bool halo_next = kc + 1 < kchunks;
#ifndef CRIMAC_EXP_NOLOAD
    if (t == 0 && halo_next) load_halo(kc + 1);
#endif
#ifndef CRIMAC_EXP_NOCOMPUTE
    compute(kc & 1, par, t);
#endif
#ifndef CRIMAC_EXP_NOSTORE
    if (s + 1 < nsteps) store_b(SetNext{}, 1 - par);
    if (t == 7 && halo_next) store_halo((kc + 1) & 1);
#endif
#ifndef CRIMAC_EXP_NOBARRIER
    __syncthreads();
#endif
    if (++t == 9) { t = 0; ++kc; }
  };
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    step(I0{}, s);
    step(I1{}, s + 1);
  }
  if (s < nsteps) step(I0{}, s);

  // ---- epilogue (conv_epilogue.h): bias/ReLU, fused reductions, LDS-staged coalesced stores -------
  conv_epilogue<TO, BN, BM, 256, MT, NT, f32x16>(acc, p.epi, smem, b, y0, x0, n0, TR, wr, wc);
}

template <typename TA, int NPL, int BN, int BK, int TR, typename P16 = typename PlaneOf<TA>::type, typename TO = TA>
int launch(ConvParams p, hipStream_t st) {
  constexpr int HALO_ROWS = (TR + 2) * HP;
  constexpr int BM = TR * TC;
  p.tiles_y = cdiv(p.H, TR);
  p.tiles_x = cdiv(p.W, TC);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x * (p.N / BN);
  size_t lds = (size_t)2 * NPL * (HALO_ROWS * BK * 2 + BN * BK * 2);
  const size_t stage = (size_t)BM * (BN * sizeof(TA) + 16) + 2 * BN * sizeof(float);
  if (stage > lds) lds = stage;      // the epilogue staging tile reuses the operand buffers
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<TA, NPL, BN, BK, TR, P16, TO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((conv3x3_kernel<TA, NPL, BN, BK, TR, P16, TO>), dim3((unsigned)ntiles), dim3(256), lds, st, p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}

}  // namespace

// conv3x3_glds.hip: 16-bit-storage direct-to-LDS variants (fp16 != 0: IEEE half, else bf16)
int crimac_conv3x3_glds_16_f32out(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w_hi,
                                  const EpiParams& epi, hipStream_t st, int n_first, int n_count, int wfrag);
int crimac_conv3x3_glds_16(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w_hi,
                           const EpiParams& epi, hipStream_t st, int n_first, int n_count, int fp16, int wfrag);
int crimac_conv3x3_c16_16(const void* in, long in_ld, int B, int H, int W, int N, const void* w_hi,
                          const EpiParams& epi, hipStream_t st, int fp16);
// conv3x3_glds.hip: plane-pair input (CRIMAC_PREC_H3P), Cin % 32 == 0; out_planes: plane-pair output, else fp32
int crimac_conv3x3_c16_hp(const void* in, long in_ld, int B, int H, int W, int N, const void* w, const EpiParams& epi,
                          hipStream_t st, int out_planes);
int crimac_conv3x3_glds_hp(const void* in, long in_ld, int B, int H, int W, int Cin, int N, const void* w,
                           const EpiParams& epi, hipStream_t st, int n_first, int n_count, int out_planes, int wfrag);

// register-staged kernel for one 16-bit storage type (odd channel counts, A/B runs)
template <typename T16>
static int conv3x3_staged16(const ConvParams& p, hipStream_t st, int Cin, bool n128, bool big, int force_bk) {
  // N = 64 layers (level 0 / decoder 3, also the HBM-heaviest): the 32-deep chunk halves the LDS
  // footprint -> 3-4 workgroups per CU, measured 10-16 % faster there; 64-deep wins for N >= 128
  if (Cin % 64 == 0 && force_bk != 32 && (n128 || force_bk == 64)) {
    if (big) return n128 ? launch<T16, 1, 128, 64, 16>(p, st) : launch<T16, 1, 64, 64, 16>(p, st);
    return n128 ? launch<T16, 1, 128, 64, 8>(p, st) : launch<T16, 1, 64, 64, 8>(p, st);
  }
  if (Cin % 32 == 0) return n128 ? launch<T16, 1, 128, 32, 8>(p, st) : launch<T16, 1, 64, 32, 8>(p, st);
  return n128 ? launch<T16, 1, 128, 16, 8>(p, st) : launch<T16, 1, 64, 16, 8>(p, st);
}

static int conv3x3_run(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                       const void* w_hi, const void* w_lo, const float* bias, void* out,
                       long out_ld, int relu, int stat_mode, double* stat_sum, double* stat_sumsq,
                       int stat_replicas, const void* bnb_y, long bnb_y_ld, const float* bnb_vec,
                       long bnb_stride, int n_first, int n_count, void* stream, void* pool_out = nullptr,
                       long pool_ld = 0) {
  if (prec == CRIMAC_PREC_H3F_BWD) {
    // the backward pass of 'h3f': fp16 operands.  With CRIMAC_EPI_OUT_PLANES the output is an MFMA operand too (the up half
    // of a decoder block's d(concat)): fp16 in, fp16 out -- the fp16 mode's launch; otherwise fp32 out
    CRIMAC_REQUIRE(!pool_out && Cin > 0 && Cin % 64 == 0 && N > 0 && N % 64 == 0 && in && w_hi && out && B > 0 && H > 0 && W > 0,
                   "conv3x3 (H3F_BWD): Cin, N multiples of 64 (got %d, %d)", Cin, N);
    if (relu & CRIMAC_EPI_OUT_PLANES)      // (column sums taken here are a bias gradient: of the unrounded results)
      return conv3x3_run(CRIMAC_PREC_FP16, in, in_ld, B, H, W, Cin, N, w_hi, w_lo, bias, out, out_ld,
                         (relu & (CRIMAC_EPI_RELU | CRIMAC_EPI_WFRAG | CRIMAC_EPI_WROWS)) | CRIMAC_EPI_STAT_RAW,
                         stat_mode, stat_sum, stat_sumsq, stat_replicas, bnb_y, bnb_y_ld, bnb_vec, bnb_stride, n_first,
                         n_count, stream);
    CRIMAC_REQUIRE(in_ld >= Cin && in_ld % 8 == 0 && out_ld >= n_first + n_count && out_ld % 8 == 0, "conv3x3: bad pixel strides (in_ld=%ld out_ld=%ld)", in_ld, out_ld);
    CRIMAC_REQUIRE(stat_mode >= 0 && stat_mode <= 2 && (stat_mode == 0 || (stat_sum && stat_sumsq && stat_replicas >= 1)),
                   "conv3x3: stat_mode %d needs both accumulators and replicas >= 1", stat_mode);
    CRIMAC_REQUIRE(stat_mode != 2 || (bnb_y && bnb_vec && bnb_y_ld >= N && bnb_y_ld % 8 == 0 && bnb_stride >= N),
                   "conv3x3: stat_mode 2 needs y, its pixel stride and the BatchNorm vectors");
    CRIMAC_REQUIRE(n_first >= 0 && n_count > 0 && n_first + n_count <= N, "conv3x3_cols: bad channel range");
    EpiParams e{};
    e.bias = bias; e.out = out; e.out_ld = out_ld; e.relu = relu & CRIMAC_EPI_RELU; e.H = H; e.W = W; e.N = N;
    e.stat_mode = stat_mode; e.stat_sum = stat_mode ? stat_sum : nullptr; e.stat_sumsq = stat_sumsq;
    e.stat_replicas = stat_replicas > 0 ? stat_replicas : 1;
    e.bnb_y = bnb_y; e.bnb_y_ld = bnb_y_ld; e.bnb_vec = bnb_vec; e.bnb_stride = bnb_stride;
    e.acc_scale = 1.f;
    return crimac_conv3x3_glds_16_f32out(in, in_ld, B, H, W, Cin, N, w_hi, e, (hipStream_t)stream, n_first, n_count,
                                         (relu & CRIMAC_EPI_WROWS) ? 2 : (relu & CRIMAC_EPI_WFRAG) != 0);
  }
  CRIMAC_REQUIRE(prec >= CRIMAC_PREC_BF16 && prec <= CRIMAC_PREC_MAX, "conv3x3: bad precision %d", prec);
  CRIMAC_REQUIRE(!pool_out || (H % 2 == 0 && W % 2 == 0 && pool_ld >= N && pool_ld % 8 == 0 && stat_mode == 0 &&
                               n_first == 0 && n_count == N && !(Cin == 16 && N == 64)),
                 "conv3x3_pool: needs even H, W, pool_ld >= N (multiple of 8), no fused reduction, the whole channel "
                 "range, and not the first-layer kernel");
  CRIMAC_REQUIRE(Cin > 0 && Cin % 16 == 0, "conv3x3: Cin=%d must be a positive multiple of 16", Cin);
  CRIMAC_REQUIRE(N > 0 && N % 64 == 0, "conv3x3: N=%d must be a positive multiple of 64", N);
  // (a channel range [n_first, +n_count) only touches the output columns below n_first + n_count: a range that starts at
  // 0 may go to a buffer of its own with a narrower pixel stride)
  CRIMAC_REQUIRE(in_ld >= Cin && in_ld % 8 == 0 && out_ld >= n_first + n_count && out_ld % 8 == 0,
                 "conv3x3: bad pixel strides (in_ld=%ld out_ld=%ld)", in_ld, out_ld);
  CRIMAC_REQUIRE(B > 0 && H > 0 && W > 0 && in && w_hi && out, "conv3x3: bad arguments");
  const bool is16 = prec == CRIMAC_PREC_BF16 || prec == CRIMAC_PREC_FP16;
  const int fp16 = prec == CRIMAC_PREC_FP16;
  CRIMAC_REQUIRE(is16 || w_lo || prec == CRIMAC_PREC_H3P, "conv3x3: split precisions need the low weight plane(s)");
  CRIMAC_REQUIRE(stat_mode >= 0 && stat_mode <= 2, "conv3x3: stat_mode=%d", stat_mode);
  CRIMAC_REQUIRE(stat_mode == 0 || (stat_sum && stat_sumsq && stat_replicas >= 1),
                 "conv3x3: stat_mode %d needs both accumulators and replicas >= 1", stat_mode);
  CRIMAC_REQUIRE(stat_mode != 2 || (bnb_y && bnb_vec && bnb_y_ld >= N && bnb_y_ld % 8 == 0 && bnb_stride >= N),
                 "conv3x3: stat_mode 2 needs y, its pixel stride and the BatchNorm vectors");
  ConvParams p;
  p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.w_hi = (const unsigned short*)w_hi; p.w_lo = (const unsigned short*)w_lo;
  EpiParams& e = p.epi;
  const int out_planes = (relu & CRIMAC_EPI_OUT_PLANES) != 0;
  const int cin4 = (relu & CRIMAC_EPI_CIN4) != 0;
  e.stat_raw = (relu & CRIMAC_EPI_STAT_RAW) != 0;
  CRIMAC_REQUIRE(!(relu & CRIMAC_EPI_WROWS) || (relu & CRIMAC_EPI_WFRAG), "conv3x3: CRIMAC_EPI_WROWS reads a fragment-major plane (CRIMAC_EPI_WFRAG)");
  const int wfrag = (relu & CRIMAC_EPI_WROWS) ? 2 : (relu & CRIMAC_EPI_WFRAG) != 0;
  relu &= CRIMAC_EPI_RELU;
  CRIMAC_REQUIRE(!out_planes || (prec == CRIMAC_PREC_H3P && stat_mode != 2),
                 "conv3x3: plane-pair output is an H3P option (never with the fused BatchNorm-backward sums)");
  p.w_il = prec == CRIMAC_PREC_H3P;
  e.bias = bias; e.out = out; e.out_ld = out_ld; e.relu = relu; e.H = H; e.W = W; e.N = N;
  e.stat_mode = stat_mode; e.stat_sum = stat_mode ? stat_sum : nullptr; e.stat_sumsq = stat_sumsq;
  e.stat_replicas = stat_replicas > 0 ? stat_replicas : 1;
  e.bnb_y = bnb_y; e.bnb_y_ld = bnb_y_ld; e.bnb_vec = bnb_vec; e.bnb_stride = bnb_stride;
  e.acc_scale = (prec == CRIMAC_PREC_F32H3 || prec == CRIMAC_PREC_H3P) ? 1.f / (float)(1 << CRIMAC_F32H3_WSHIFT) : 1.f;
  e.pool_out = pool_out; e.pool_ld = pool_ld;
  hipStream_t st = (hipStream_t)stream;
  const bool n128 = N % 128 == 0;
  // measured (tools/bench_conv.py): the 256-pixel tile of THIS kernel (4 waves, 1 workgroup/CU) loses
  // 5-20 % to the 128-pixel tile (2 workgroups/CU) on every layer -- occupancy beats LDS traffic here
  static const int force_tr = getenv("CRIMAC_CONV_TR") ? atoi(getenv("CRIMAC_CONV_TR")) : 0;
  const bool big = force_tr == 16;
  static const int force_bk = getenv("CRIMAC_CONV_BK") ? atoi(getenv("CRIMAC_CONV_BK")) : 0;
  // bf16 with 64-deep channel chunks: LDS-DMA streaming kernels (CRIMAC_CONV_GLDS=0 selects the
  // register-staged kernel below, kept for A/B measurements and as the fp32-mode structure).
  // Measured (tools/bench_conv.py, B=32): +8-18 % on every layer with N >= 128 (8-wave kernel), +5-13 % on
  // the N = 64 layers (4-wave kernel, two workgroups per CU).
  static const int use_glds = getenv("CRIMAC_CONV_GLDS") ? atoi(getenv("CRIMAC_CONV_GLDS")) : 1;
  const bool ranged = n_first != 0 || n_count != N;
  CRIMAC_REQUIRE(!ranged || (n_first >= 0 && n_count > 0 && n_first + n_count <= N && use_glds &&
                             ((is16 && Cin % 64 == 0) || (prec == CRIMAC_PREC_H3P && Cin % 32 == 0))),
                 "conv3x3_cols: a channel range needs the LDS-DMA kernels (16-bit storage: Cin %% 64 == 0, plane pairs: "
                 "Cin %% 32 == 0)");
  if (prec == CRIMAC_PREC_H3P) {
    // plane-pair input: the 16-bit LDS-DMA kernels on a tensor of 2 Cin halves per pixel (3 MFMAs per product);
    // the first layer (4 input channels padded to 16) runs on the register-staged kernel, whose staging copies the
    // pre-split halves
    CRIMAC_REQUIRE(!wfrag || (Cin % 32 == 0 && use_glds),
                   "conv3x3 (plane pairs): fragment-major weights (CRIMAC_EPI_WFRAG) are read by the channel-split kernel only");
    if (Cin % 32 == 0 && use_glds)
      return crimac_conv3x3_glds_hp(in, in_ld, B, H, W, Cin, N, w_hi, e, st, n_first, n_count, out_planes, wfrag);
    CRIMAC_REQUIRE(!pool_out, "conv3x3_pool (plane pairs): Cin %% 32 == 0 only");
    // first layer (4 input channels padded to 16): the persistent 16-channel kernel on hi / lo pseudo-channels
    static const int c16hp = getenv("CRIMAC_CONV_C16HP") ? atoi(getenv("CRIMAC_CONV_C16HP")) : 1;
    if (c16hp && cin4 && Cin == 16 && N == 64 && use_glds && stat_mode != 2 && in_ld >= 8)
      return crimac_conv3x3_c16_hp(in, in_ld, B, H, W, N, w_hi, e, st, out_planes);
    if (Cin % 32 == 0) {
      if (out_planes) return n128 ? launch<hp_t, 2, 128, 32, 8, half_t, hp_t>(p, st) : launch<hp_t, 2, 64, 32, 8, half_t, hp_t>(p, st);
      return n128 ? launch<hp_t, 2, 128, 32, 8, half_t, float>(p, st) : launch<hp_t, 2, 64, 32, 8, half_t, float>(p, st);
    }
    if (out_planes) return n128 ? launch<hp_t, 2, 128, 16, 8, half_t, hp_t>(p, st) : launch<hp_t, 2, 64, 16, 8, half_t, hp_t>(p, st);
    return n128 ? launch<hp_t, 2, 128, 16, 8, half_t, float>(p, st) : launch<hp_t, 2, 64, 16, 8, half_t, float>(p, st);
  }
  CRIMAC_REQUIRE(!wfrag || (is16 && Cin % 64 == 0 && use_glds),
                 "conv3x3: fragment-major weights (CRIMAC_EPI_WFRAG) are read by the 16-bit / plane-pair channel-split kernel only");
  if (is16 && Cin % 64 == 0 && use_glds)
    return crimac_conv3x3_glds_16(in, in_ld, B, H, W, Cin, N, w_hi, e, st, n_first, n_count, fp16, wfrag);
  // first layer (4 input channels padded to 16): persistent one-barrier kernel
  if (is16 && Cin == 16 && N == 64 && use_glds)
    return crimac_conv3x3_c16_16(in, in_ld, B, H, W, N, w_hi, e, st, fp16);
  if (is16)
    return fp16 ? conv3x3_staged16<half_t>(p, st, Cin, n128, big, force_bk)
                : conv3x3_staged16<bf16_t>(p, st, Cin, n128, big, force_bk);
  // split planes keep 2-3 planes per operand: 32-deep chunks keep the LDS footprint in bounds
  if (prec == CRIMAC_PREC_F32H3) {       // two fp16 planes (forward operands only: see the header)
    if (Cin % 32 == 0) return n128 ? launch<float, 2, 128, 32, 8, half_t>(p, st) : launch<float, 2, 64, 32, 8, half_t>(p, st);
    return n128 ? launch<float, 2, 128, 16, 8, half_t>(p, st) : launch<float, 2, 64, 16, 8, half_t>(p, st);
  }
  if (prec == CRIMAC_PREC_F32X3) {
    if (Cin % 32 == 0) return n128 ? launch<float, 2, 128, 32, 8>(p, st) : launch<float, 2, 64, 32, 8>(p, st);
    return n128 ? launch<float, 2, 128, 16, 8>(p, st) : launch<float, 2, 64, 16, 8>(p, st);
  }
  if (Cin % 32 == 0) return n128 ? launch<float, 3, 128, 32, 8>(p, st) : launch<float, 3, 64, 32, 8>(p, st);
  return n128 ? launch<float, 3, 128, 16, 8>(p, st) : launch<float, 3, 64, 16, 8>(p, st);
}

extern "C" int crimac_conv3x3(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                              const void* w_hi, const void* w_lo, const float* bias, void* out,
                              long out_ld, int relu, int stat_mode, double* stat_sum, double* stat_sumsq,
                              int stat_replicas, const void* bnb_y, long bnb_y_ld, const float* bnb_vec,
                              long bnb_stride, void* stream) {
  return conv3x3_run(prec, in, in_ld, B, H, W, Cin, N, w_hi, w_lo, bias, out, out_ld, relu, stat_mode, stat_sum,
                     stat_sumsq, stat_replicas, bnb_y, bnb_y_ld, bnb_vec, bnb_stride, 0, N, stream);
}

// The same convolution restricted to the output channels [n_first, n_first + n_count): every pointer (weights,
// bias, out, accumulators) is that of the FULL N-channel convolution.  Lets the input gradient of a decoder
// block's first convolution be produced in two launches -- the half that the up-convolution's backward needs at
// once, and the skip-connection half, which is not needed until the encoder level is reached (side stream).
extern "C" int crimac_conv3x3_cols(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                                   const void* w_hi, const void* w_lo, const float* bias, void* out,
                                   long out_ld, int relu, int stat_mode, double* stat_sum, double* stat_sumsq,
                                   int stat_replicas, const void* bnb_y, long bnb_y_ld, const float* bnb_vec,
                                   long bnb_stride, int n_first, int n_count, void* stream) {
  return conv3x3_run(prec, in, in_ld, B, H, W, Cin, N, w_hi, w_lo, bias, out, out_ld, relu, stat_mode, stat_sum,
                     stat_sumsq, stat_replicas, bnb_y, bnb_y_ld, bnb_vec, bnb_stride, n_first, n_count, stream);
}

// Eval-mode encoder block tail (conv3x3 + folded BatchNorm + ReLU, then nn.MaxPool2d(2, 2), unet.py:85-92): the same
// convolution that ALSO writes the 2x2/2 max-pool of its stored output -- the pooled tensor comes out of the conv
// epilogue (the tile is still in LDS) instead of a second pass over the skip tensor.
extern "C" int crimac_conv3x3_pool(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                                   const void* w_hi, const void* w_lo, const float* bias, void* out, long out_ld,
                                   int relu, void* pool_out, long pool_ld, void* stream) {
  CRIMAC_REQUIRE(pool_out, "conv3x3_pool: pool_out is NULL (use crimac_conv3x3)");
  return conv3x3_run(prec, in, in_ld, B, H, W, Cin, N, w_hi, w_lo, bias, out, out_ld, relu, 0, nullptr, nullptr, 1,
                     nullptr, 0, nullptr, 0, 0, N, stream, pool_out, pool_ld);
}
