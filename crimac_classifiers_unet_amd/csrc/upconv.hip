// ConvTranspose2d(k=2, s=2) forward and input gradient on MFMA for CDNA4 (gfx950), bf16, NHWC.
//
// Reference op: nn.ConvTranspose2d(k2, s2) (unet.py:47-49, :130): four independent 1x1 GEMMs, no overlap-add
// (SURVEY.md A2).  Both directions are plain GEMMs over a 16x16 tile of COARSE pixels:
//   forward : C[m][(ab, co)] = sum_ci X[m][ci] * W[ci][co][ab]            K = Cin,      N = 4*Cout, scatter
//   dgrad   : C[m][ci]       = sum_ab sum_co dY[fine(m, ab)][co] * W[ci][co][ab]   K = 4*Cout, N = Cin
// Same structure as the channel-split 3x3 kernel (conv3x3_glds.hip, conv3x3_wch_kernel): 4 waves, each owns
// 32 output columns of the whole 256-pixel tile and takes its weight fragments straight from global memory
// into registers; the A operand (256 pixel rows x 64 channels = 32 KB per K-chunk) arrives by LDS-DMA through
// a buffer resource (per-lane 32-bit offsets; the tap / chunk offset is a scalar; out-of-image lanes read out
// of range and get zeros), double-buffered: chunk c+1 is requested when chunk c starts.  One barrier per
// chunk (64 MFMAs of 16x16x32 per wave), fragment reads one group ahead of the MFMAs (inline asm, counted lgkmcnt).
// Two workgroups per CU (70.6 KB LDS: the epilogue staging tile).  Replaces the register-staged igemm kernel
// for these two ops (0.43 -> see DESIGN.md PFLOP/s); igemm remains for the fp32 parity modes and odd shapes.
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int TP = 16;                 // tile: TP x TP coarse pixels
constexpr int BM = TP * TP;            // 256 GEMM rows
constexpr int BN = 128, BK = 64, RB = BK * 2;
constexpr int A_BYTES = BM * RB;       // 32 KB per chunk
constexpr int NW = 4;
constexpr int NA = A_BYTES / 1024 / NW;   // 8 DMA wave-instructions per wave and chunk

struct UpParams {
  const void* in;
  long in_ld;
  int B, H, W;          // COARSE grid
  int K;                // channels per tap of the A operand (forward: Cin; dgrad: Cout)
  int N;                // GEMM columns (forward: 4*Cout; dgrad: Cin)
  int ntaps;            // 1 (forward) or 4 (dgrad: A row of tap (a,b) = fine pixel (2y+a, 2x+b))
  const unsigned short* w;   // [ntaps][N][K] bf16
  const float* bias;    // forward: [Cout]
  int cout;             // forward: columns per (a,b) group
  void* out;
  long out_ld;
  EpiParams epi;        // dgrad epilogue (dense tile)
  int tiles_y, tiles_x;
};

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128_asm(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// MFMA shape 16x16x32 (16 x 2 tiles per wave; the 32x32x16 form is ~8 % slower, see conv3x3_glds.hip): a chunk
// is 8 groups of 4 fragment reads + 8 MFMAs, group H = (k-step H / 4, M tiles 4 * (H % 4) .. + 3)
struct Frags {
  bf16x8 a[2][4];        // two groups in flight
  bf16x8 b[2][4];        // [0]: this chunk's weight fragments [ks * 2 + nb], [1]: landing zone of the next chunk's
};
template <int H>
__device__ __forceinline__ void issue_half(const unsigned (&av)[2], Frags& f) {
  constexpr int ks = H / 4, q = H % 4;
  f.a[H & 1][0] = lds_read128_asm<(q * 4 + 0) * 16 * RB>(av[ks]);
  f.a[H & 1][1] = lds_read128_asm<(q * 4 + 1) * 16 * RB>(av[ks]);
  f.a[H & 1][2] = lds_read128_asm<(q * 4 + 2) * 16 * RB>(av[ks]);
  f.a[H & 1][3] = lds_read128_asm<(q * 4 + 3) * 16 * RB>(av[ks]);
}
template <bool LAST>
__device__ __forceinline__ void release_half(Frags& f, int set) {
  if constexpr (LAST)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[set][0]), "+v"(f.a[set][1]), "+v"(f.a[set][2]), "+v"(f.a[set][3])
                 :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.a[set][0]), "+v"(f.a[set][1]), "+v"(f.a[set][2]), "+v"(f.a[set][3])
                 :: "memory");
}
// PP (plane pairs, common.h hp_t): k-step 0 of a chunk = hi plane of 32 channels, k-step 1 = their lo plane, in both
// operands -> hi*lo + hi*hi on the first k-step's fragments, lo*hi on the second's (3 MFMAs per fragment pair)
template <typename T16, bool PP, int H, typename ACC>
__device__ __forceinline__ void half(const unsigned (&av)[2], Frags& f, ACC& acc) {
  constexpr int ks = H / 4, q = H % 4;
  if constexpr (H + 1 < 8) {
    issue_half<H + 1>(av, f);
    release_half<false>(f, H & 1);
  } else {
    release_half<true>(f, H & 1);
  }
  if constexpr (PP && ks == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        acc[q * 4 + j][nb] = E16<T16>::mfma16(f.a[H & 1][j], f.b[0][2 + nb], acc[q * 4 + j][nb]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
      acc[q * 4 + j][nb] = E16<T16>::mfma16(f.a[H & 1][j], f.b[0][(PP ? 0 : ks * 2) + nb], acc[q * 4 + j][nb]);
  if constexpr (H + 1 < 8) half<T16, PP, H + 1>(av, f, acc);
}
// s0: row of the wave's first 16 columns, s1: of the second 16; fragments [ks * 2 + nb]
__device__ __forceinline__ void load_b(const unsigned short* s0, const unsigned short* s1, bf16x8 (&bf)[4]) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\t"
      "global_load_dwordx4 %1, %5, off\n\t"
      "global_load_dwordx4 %2, %4, off offset:64\n\t"
      "global_load_dwordx4 %3, %5, off offset:64"
      : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2]), "=&v"(bf[3])
      : "v"(s0), "v"(s1)
      : "memory");
}

// Epilogue staging of an output storage type (as in conv3x3_glds.hip): fp32 / plane pairs in two 128-row slices
template <typename TO> struct EpiPasses {
  static constexpr int kStageBytes = sizeof(TO) == 2 ? 2 : 4;
  static constexpr int value = sizeof(TO) == 2 ? 1 : 2;
};
template <bool PP> __device__ __forceinline__ int src_unit(int u) { return PP ? (((u & 3) << 1) | (u >> 2)) : u; }

// SCATTER: forward (output on the fine grid, column group (a,b) -> pixel (2y+a, 2x+b)); else dense dgrad tile.
// PP: plane-pair input (p.K / p.in_ld count halves); TO: output storage type (T16, float, or hp_t plane pairs)
// NWV: waves per workgroup = 32-column groups sharing the A tile: 4 (128 columns, two workgroups per CU) or 8 (256 columns,
// one workgroup per CU: an experiment, see launch()).  Every column group of a pixel tile stages the same A chunks
// (plane pairs, 1024 -> 512 forward: 16 column groups x 33.5 MB = 0.54 GB through the LDS-DMA path in 105 us).
template <bool SCATTER, typename T16, typename TO = T16, bool PP = false, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2, 2)))
void upconv_wch_kernel(UpParams p) {
  constexpr int BN = 32 * NWV, NW = NWV, NA = A_BYTES / 1024 / NWV, NTHR = 64 * NWV;      // (shadow the 4-wave constants)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid of (pixel tiles) x (column groups).  Workgroup id -> XCD id % 8; an XCD owns a contiguous range of pixel tiles
  // and walks it with the COLUMN GROUP fastest: the ncol workgroups that stage the same A tile are dispatched back to
  // back on one XCD, so all but the first find it in that XCD's L2 (column-group-major order re-read the tile from HBM
  // ncol times: 128 -> 64 channels at 128^2 moved 536 MB for 402 MB of tensors).
  const int ncol = p.N / BN;
  int bid, colg;
  {
    const int nt = (int)(gridDim.x / ncol), q = nt / 8, r = nt % 8, x = blockIdx.x % 8, j = blockIdx.x / 8;
    // XCD x owns q (+1 for x < r) pixel tiles = that many times ncol workgroups; ids beyond an XCD's share (when r != 0 the
    // shares differ by ncol) cannot occur: gridDim.x = nt * ncol and id % 8 walks the XCDs evenly only if shares are equal,
    // so uneven tile counts fall back to the plain order below
    if (r == 0) {
      colg = j % ncol;
      bid = x * q + j / ncol;
    } else {
      colg = (int)(blockIdx.x % ncol);
      bid = (int)(blockIdx.x / ncol);
    }
  }
  int tile_m = bid;
  const int txi = tile_m % p.tiles_x;
  tile_m /= p.tiles_x;
  const int tyi = tile_m % p.tiles_y;
  const int b = tile_m / p.tiles_y;
  const int y0 = tyi * TP, x0 = txi * TP;
  const int n0 = colg * BN;

  // ---- A operand: lane -> (tile row, 16-byte unit); source unit = unit ^ swizzle(row) ----------------
  // forward: row r reads coarse pixel (y, x); dgrad: fine pixel (2y + a, 2x + b) of dY, the tap's (a, b) part of
  // the offset is the same for every lane and rides in the scalar offset with the K-chunk
  const int Hi = SCATTER ? p.H : 2 * p.H, Wi = SCATTER ? p.W : 2 * p.W;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.in), 0, (int)((((long)p.B * Hi * Wi - 1) * p.in_ld + p.K) * 2), 0x00020000);
  unsigned voff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    const int y = y0 + (row >> 4), x = x0 + (row & 15);
    const int us = src_unit<PP>((lane & 7) ^ ((row >> 1) & 7));
    const long pix = SCATTER ? ((long)b * Hi + y) * Wi + x : ((long)b * Hi + 2 * y) * Wi + 2 * x;
    voff[i] = (y < p.H && x < p.W) ? (unsigned)((pix * p.in_ld + us * 8) * 2) : 0x80000000u;
  }
  const int kchunks = p.K / BK, nchunks = p.ntaps * kchunks;
  auto chunk_soff = [&](int c) {          // byte offset of chunk c = (tap, k-chunk) relative to tap 0, chunk 0
    const int t = c / kchunks, kc = c - t * kchunks;
    return (int)((((long)(t >> 1) * Wi + (t & 1)) * p.in_ld + kc * BK) * 2);
  };
  auto issue_a = [&](int c, int buf) {
    const int soff = chunk_soff(c);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(smem + buf * A_BYTES + (wave + NW * i) * 1024), 16,
          (int)voff[i], soff, 0, 0);
  };

  // ---- B operand: this wave's 32 columns, rows n (+16) of w[tap][N][K], k = kc*64 + ks*32 + fq*8 -----
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned short* wrow = p.w + (long)(n0 + 32 * wave + fr) * p.K + fq * 8;
  const long w_tap = (long)p.N * p.K, w_nb = 16L * p.K;
  auto b_src = [&](int c) {
    const int t = c / kchunks, kc = c - t * kchunks;
    return wrow + t * w_tap + kc * BK;
  };

  unsigned av0[2];
  {
    const unsigned a_lds = (unsigned)(unsigned long)((LDS_PTR(unsigned char))(smem));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) av0[ks] = a_lds + fr * RB + (((4 * ks + fq) ^ ((fr >> 1) & 7)) << 4);
  }

  f32x4 acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // NOTE on the asm loads: between a load and the wait that lands it the compiler believes the destination
  // registers hold their values and may copy them (at a loop back-edge, or to resolve a phi after a branch) --
  // so a load is issued unconditionally and landed (waited for and tied) in the same straight-line code.
  Frags f;
  load_b(b_src(0), b_src(0) + w_nb, f.b[1]);
  issue_a(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[1][2]), "+v"(f.b[1][3]) :: "memory");
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    __builtin_amdgcn_s_barrier();        // chunk c landed for every wave; the other buffer is no longer read
#pragma unroll
    for (int k = 0; k < 4; ++k) f.b[0][k] = f.b[1][k];
    // the weight load is UNCONDITIONAL (the last chunk re-requests its own fragments): a load inside a branch
    // makes its destination a phi, and the copies that resolve it would read the registers while in flight
    {
      const unsigned short* s = b_src(c + 1 < nchunks ? c + 1 : c);
      load_b(s, s + w_nb, f.b[1]);
    }
    if (c + 1 < nchunks) issue_a(c + 1, buf ^ 1);
    unsigned av[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) av[ks] = av0[ks] + buf * A_BYTES;
    issue_half<0>(av, f);
    half<T16, PP, 0>(av, f, acc);
    // chunk c+1's A tile and weight fragments were requested 64 MFMAs ago
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[1][2]), "+v"(f.b[1][3]) :: "memory");
  }
  __syncthreads();                       // staging below reuses the A buffers

  if constexpr (!SCATTER) {
    conv_epilogue<TO, BN, BM, NTHR, 16, 2, f32x4, -1, EpiPasses<TO>::value>(acc, p.epi, smem, b, y0, x0, n0, TP, 0, wave);
  } else {
    // bias, then the tile through LDS: [256 rows][128 cols + pad] (16-bit types: all rows at once; fp32 / plane pairs:
    // two slices of 128 rows, staged as fp32), then 16-byte stores scattered to the fine grid: column
    // n = (a*2 + bb) * cout + co -> pixel (2y + a, 2x + bb), channel co
    constexpr bool HPO = __is_same(TO, hp_t);
    using TS = typename std::conditional<HPO, float, TO>::type;
    constexpr int PASSES = EpiPasses<TO>::value, RPASS = BM / PASSES;
    constexpr int PITCH = BN * (int)sizeof(TS) + 16;
    TO* outp = reinterpret_cast<TO*>(p.out);
    // 16-bit outputs: 16 lanes x 8 columns per row, 16 rows per round.  4-byte outputs (fp32 / plane pairs): 32 lanes
    // x 16 BYTES per row, 8 rows per round -- consecutive lanes on consecutive 16 bytes (whole lines per wave-instruction,
    // conv_epilogue.h); plane pairs: the even lane of a pair stores the hi plane of an 8-column group, the odd lane the lo
    constexpr bool W4 = sizeof(TS) == 4;
    constexpr int LPR = W4 ? BN / 4 : BN / 8, RPR = NTHR / LPR;       // lanes per row; rows per round: 8 (4-byte), 16 (16-bit)
    const int cl = tid % LPR, r0 = tid / LPR;
    const int n = n0 + (W4 ? (HPO ? (cl >> 1) * 8 : cl * 4) : cl * 8);
    const int ab = n / p.cout, co = n - ab * p.cout;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if (ps > 0) __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int col = wave * 32 + nb * 16 + (lane & 15);
        const float bv = p.bias ? p.bias[(n0 + col) % p.cout] : 0.f;
#pragma unroll
        for (int i = ps * (16 / PASSES); i < (ps + 1) * (16 / PASSES); ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + (lane >> 4) * 4 + r - ps * RPASS;
            *reinterpret_cast<TS*>(smem + row * PITCH + col * (int)sizeof(TS)) = (TS)(acc[i][nb][r] * p.epi.acc_scale + bv);
          }
      }
      __syncthreads();
#pragma unroll(HPO ? 4 : RPASS / RPR)
      for (int rr = 0; rr < RPASS / RPR; ++rr) {
        const int lrow = r0 + rr * RPR, row = ps * RPASS + lrow;
        const int y = y0 + (row >> 4), x = x0 + (row & 15);
        if (y < p.H && x < p.W) {
          const long pix = ((long)b * 2 * p.H + 2 * y + (ab >> 1)) * (2 * p.W) + 2 * x + (ab & 1);
          if constexpr (HPO) {
            float v[8];
            load8(reinterpret_cast<const float*>(smem + lrow * PITCH) + (cl >> 1) * 8, v);
            u32x4 hi, lo;
            hp_split(v, hi, lo);
            *(reinterpret_cast<u32x4*>(outp + pix * p.out_ld + co) + (cl & 1)) = (cl & 1) ? lo : hi;
          } else if constexpr (W4) {
            *reinterpret_cast<u32x4*>(outp + pix * p.out_ld + co) =
                *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(smem + lrow * PITCH) + cl * 4);
          } else {
            *reinterpret_cast<u32x4*>(outp + pix * p.out_ld + co) =
                *reinterpret_cast<const u32x4*>(reinterpret_cast<const TS*>(smem + lrow * PITCH) + cl * 8);
          }
        }
      }
    }
  }
}

template <bool SCATTER, typename T16, typename TO, bool PP, int NWV>
int launch_n(UpParams p, hipStream_t st) {
  constexpr int BNK = 32 * NWV;
  p.tiles_y = cdiv(p.H, TP);
  p.tiles_x = cdiv(p.W, TP);
  const long ntiles = (long)p.B * p.tiles_y * p.tiles_x;
  constexpr size_t stage = (size_t)(BM / EpiPasses<TO>::value) * (BNK * EpiPasses<TO>::kStageBytes + 16) + 2 * BNK * 4;
  static_assert(stage <= 160 * 1024, "epilogue staging");
  const size_t lds = stage > 2 * (size_t)A_BYTES ? stage : 2 * (size_t)A_BYTES;
  static unsigned long long attr_devs = 0;      // bit d: done on device d (the attribute is per device)
  if (crimac_first_use_on_device(&attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_wch_kernel<SCATTER, T16, TO, PP, NWV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((upconv_wch_kernel<SCATTER, T16, TO, PP, NWV>), dim3((unsigned)(ntiles * (p.N / BNK))), dim3(64 * NWV), lds, st,
                     p);
  CRIMAC_LAUNCH_CHECK();
  return CRIMAC_OK;
}
// CRIMAC_UPCONV_W8=1: 256 columns per workgroup where the column count allows it.  Measured (B = 32, same box): it halves
// the A-tile traffic, but one 8-wave workgroup per CU has no second workgroup to cover its chunk barriers and epilogue:
// plane pairs 25.56 vs 25.32 ms per step (inference +0.6 %), bf16 equal (inference -2 %) -- off by default.
template <bool SCATTER, typename T16, typename TO = T16, bool PP = false>
int launch(UpParams p, hipStream_t st) {
  static const int w8 = getenv("CRIMAC_UPCONV_W8") ? atoi(getenv("CRIMAC_UPCONV_W8")) : 0;
  if (w8 && p.N % 256 == 0) return launch_n<SCATTER, T16, TO, PP, 8>(p, st);
  return launch_n<SCATTER, T16, TO, PP, 4>(p, st);
}

}  // namespace

// Called by crimac_igemm_conv (igemm.hip) for the bf16 transposed-convolution shapes this kernel covers.
// forward: in [B][H][W][K=Cin], w [1][N=4*Cout][K], out [B][2H][2W] (out_ld), bias [Cout]
// dgrad  : in = dY [B][2H][2W][K=Cout], w [4][N=Cin][K], out [B][H][W] (out_ld)
bool crimac_upconv_wch_ok(int ntaps, long in_bytes, int K, int N, int cout_up, long out_ld) {
  if (in_bytes >= (1L << 31) || K % BK != 0 || N % BN != 0 || out_ld % 8 != 0) return false;
  if (ntaps == 1) return cout_up % 32 == 0 && cout_up % 8 == 0;
  return ntaps == 4;
}

// fmt: 0 bf16, 1 fp16, 2 plane pairs in / plane pairs out, 3 plane pairs in / fp32 out (CRIMAC_PREC_H3P; K, in_ld in
// ELEMENTS of the plane-pair tensor: the kernel sees twice as many halves), 4 fp16 in / fp32 out (CRIMAC_PREC_H3F_BWD)
static int upconv_run(int ntaps, const void* in, long in_ld, int B, int H, int W, int K, int N, const void* w,
                      const float* bias, int cout_up, void* out, long out_ld, const EpiParams* bnb, hipStream_t st,
                      int fmt) {
  const int fp16 = fmt == 1;
  if (fmt == 2 || fmt == 3) { K *= 2; in_ld *= 2; }          // (plane-pair input: twice as many halves)
  UpParams p;
  p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.K = K; p.N = N; p.ntaps = ntaps;
  p.w = (const unsigned short*)w; p.bias = bias; p.cout = cout_up; p.out = out; p.out_ld = out_ld;
  p.epi = EpiParams{};
  p.epi.acc_scale = (fmt == 2 || fmt == 3) ? 1.f / (float)(1 << CRIMAC_F32H3_WSHIFT) : 1.f;
  p.epi.bias = nullptr; p.epi.out = out; p.epi.out_ld = out_ld; p.epi.relu = 0; p.epi.H = H; p.epi.W = W; p.epi.N = N;
  p.epi.stat_sum = nullptr; p.epi.stat_sumsq = nullptr; p.epi.stat_replicas = 1; p.epi.stat_mode = 0;
  if (bnb) {
    p.epi.stat_mode = 2; p.epi.stat_sum = bnb->stat_sum; p.epi.stat_sumsq = bnb->stat_sumsq;
    p.epi.stat_replicas = bnb->stat_replicas; p.epi.bnb_y = bnb->bnb_y; p.epi.bnb_y_ld = bnb->bnb_y_ld;
    p.epi.bnb_vec = bnb->bnb_vec; p.epi.bnb_stride = bnb->bnb_stride;
  }
  if (fmt == 2) {            // (plane-pair OUTPUT: the forward pass only; an input gradient is read by elementwise kernels)
    CRIMAC_REQUIRE(ntaps == 1, "upconv (plane pairs): the input gradient is stored as fp32");
    return launch<true, half_t, hp_t, true>(p, st);
  }
  if (fmt == 3) return ntaps == 1 ? launch<true, half_t, float, true>(p, st) : launch<false, half_t, float, true>(p, st);
  if (fmt == 4) {
    CRIMAC_REQUIRE(ntaps == 4, "upconv (fp16 operands, fp32 output): the input gradient only");
    return launch<false, half_t, float, false>(p, st);
  }
  if (fp16) return ntaps == 1 ? launch<true, half_t>(p, st) : launch<false, half_t>(p, st);
  return ntaps == 1 ? launch<true, bf16_t>(p, st) : launch<false, bf16_t>(p, st);
}

// plane-pair input (CRIMAC_PREC_H3P): K % 32 == 0, N % 128 == 0, tensor below 2 GiB
bool crimac_upconv_wch_hp_ok(int ntaps, long in_bytes, int K, int N, int cout_up, long out_ld) {
  if (in_bytes >= (1L << 31) || K % 32 != 0 || N % BN != 0 || out_ld % 8 != 0) return false;
  if (ntaps == 1) return cout_up % 32 == 0;
  return ntaps == 4;
}
int crimac_upconv_wch_hp(int ntaps, const void* in, long in_ld, int B, int H, int W, int K, int N, const void* w,
                         const float* bias, int cout_up, void* out, long out_ld, hipStream_t st, int out_planes) {
  return upconv_run(ntaps, in, in_ld, B, H, W, K, N, w, bias, cout_up, out, out_ld, nullptr, st, out_planes ? 2 : 3);
}

int crimac_upconv_wch_16(int ntaps, const void* in, long in_ld, int B, int H, int W, int K, int N, const void* w,
                         const float* bias, int cout_up, void* out, long out_ld, hipStream_t st, int fp16) {
  return upconv_run(ntaps, in, in_ld, B, H, W, K, N, w, bias, cout_up, out, out_ld, nullptr, st, fp16);
}

// Input gradient of the transposed convolution whose result is the `da` of a BatchNorm+ReLU block: that
// block's backward sums (sum dz, sum dz*xhat; conv_epilogue.h stat_mode 2) are taken in the epilogue, which
// removes a bn_bwd_reduce pass over (da, y) -- 4 launches, 0.21 ms per step at B = 32.
extern "C" int crimac_upconv2x2_dgrad_bnb_prec(int prec, const void* dy, long dy_ld, int B, int H, int W, int Cout, int Cin,
                                          const void* w_dg_hi, void* dx, long dx_ld, const void* bnb_y,
                                          long bnb_y_ld, const float* bnb_vec, long bnb_stride, double* stat_sum,
                                          double* stat_sumsq, int stat_replicas, void* stream) {
  CRIMAC_REQUIRE(prec == CRIMAC_PREC_BF16 || prec == CRIMAC_PREC_FP16 || prec == CRIMAC_PREC_H3P || prec == CRIMAC_PREC_H3F_BWD,
                 "upconv2x2_dgrad_bnb: 16-bit storage modes, plane pairs and H3F_BWD only (prec=%d)", prec);
  const bool hp = prec == CRIMAC_PREC_H3P;
  CRIMAC_REQUIRE(dy && w_dg_hi && dx && B > 0 && H > 0 && W > 0, "upconv2x2_dgrad_bnb: bad arguments");
  CRIMAC_REQUIRE(dy_ld >= Cout && dy_ld % 8 == 0 && dx_ld >= Cin && dx_ld % 8 == 0,
                 "upconv2x2_dgrad_bnb: bad pixel strides (dy_ld=%ld dx_ld=%ld)", dy_ld, dx_ld);
  const long in_bytes = (((long)B * 2 * H * 2 * W - 1) * dy_ld + Cout) * (hp ? 4 : 2);
  CRIMAC_REQUIRE(Cout > 0 && Cin > 0 && (hp ? crimac_upconv_wch_hp_ok(4, in_bytes, Cout, Cin, 0, dx_ld)
                                            : crimac_upconv_wch_ok(4, in_bytes, Cout, Cin, 0, dx_ld)),
                 "upconv2x2_dgrad_bnb: needs Cout %% 64 == 0, Cin %% 128 == 0 and a gradient tensor below 2 GiB "
                 "(got Cout=%d Cin=%d); use crimac_igemm_conv + crimac_bn_bwd_reduce", Cout, Cin);
  CRIMAC_REQUIRE(bnb_y && bnb_vec && stat_sum && stat_sumsq && stat_replicas >= 1 && bnb_y_ld >= Cin &&
                     bnb_y_ld % 8 == 0 && bnb_stride >= Cin,
                 "upconv2x2_dgrad_bnb: bad arguments of the fused BatchNorm-backward sums");
  EpiParams e{};
  e.stat_sum = stat_sum; e.stat_sumsq = stat_sumsq; e.stat_replicas = stat_replicas;
  e.bnb_y = bnb_y; e.bnb_y_ld = bnb_y_ld; e.bnb_vec = bnb_vec; e.bnb_stride = bnb_stride;
  return upconv_run(4, dy, dy_ld, B, H, W, Cout, Cin, w_dg_hi, nullptr, 0, dx, dx_ld, &e, (hipStream_t)stream,
                    hp ? 3 : (prec == CRIMAC_PREC_H3F_BWD ? 4 : (prec == CRIMAC_PREC_FP16 ? 1 : 0)));
}

// bf16 form (kept for existing callers)
extern "C" int crimac_upconv2x2_dgrad_bnb(const void* dy, long dy_ld, int B, int H, int W, int Cout, int Cin,
                                          const void* w_dg_hi, void* dx, long dx_ld, const void* bnb_y,
                                          long bnb_y_ld, const float* bnb_vec, long bnb_stride, double* stat_sum,
                                          double* stat_sumsq, int stat_replicas, void* stream) {
  return crimac_upconv2x2_dgrad_bnb_prec(CRIMAC_PREC_BF16, dy, dy_ld, B, H, W, Cout, Cin, w_dg_hi, dx, dx_ld, bnb_y,
                                         bnb_y_ld, bnb_vec, bnb_stride, stat_sum, stat_sumsq, stat_replicas, stream);
}
