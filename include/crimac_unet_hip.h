/*
 * crimac_unet_hip.h -- C ABI of the MI355X (gfx950) U-Net hot-path library
 *                      (libcrimac_unet_hip.so, built from crimac_classifiers_unet_amd/csrc/).
 *
 * The reference (CRIMAC-classifiers-unet) is pure Python and has NO FFI/plugin layer: its hot path
 * dispatches stock torch.nn modules to ATen.  The entry points below are what a
 * binding for that path binds instead of the ATen kernels; each one cites the reference call site
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would
 * add.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer owned by the caller (workspaces included); the library
 *     allocates nothing; one process per GPU, re-entrant per stream.  Process state it does keep: the
 *     thread-local last-error text; one-time per-DEVICE kernel attribute setup (dynamic-LDS limits, CU count;
 *     keyed on hipGetDevice, so several GPUs in one process work); and A/B tuning switches read ONCE from the
 *     environment at first use and constant afterwards (CRIMAC_CONV_W4, CRIMAC_CONV_P64, CRIMAC_CONV_TR,
 *     CRIMAC_CONV_BK, CRIMAC_CONV_GLDS, CRIMAC_UPCONV_WCH, CRIMAC_WGRAD_BLOCKS, CRIMAC_BNB_STREAM -- kernel-selection experiments
 *     only; results do not depend on them beyond summation order).  Reductions that use floating-point
 *     atomics (BatchNorm sums, bias gradients) are not bit-reproducible run to run; everything on the
 *     inference path is
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is asynchronous
 *   - return value: 0 (CRIMAC_OK) or a negative code; crimac_last_error() gives the thread-local text
 *   - activations are NHWC; `ld` arguments are the pixel stride in ELEMENTS (>= channels), so a
 *     tensor may live in a channel slice of a wider buffer (this is how torch.cat((up, skip), 1),
 *     unet.py:132, is made free)
 *   - `prec` selects the storage/compute type of activations:
 *       CRIMAC_PREC_BF16  : bf16 activations, bf16 MFMA, fp32 accumulate (throughput mode)
 *       CRIMAC_PREC_F32X3 : fp32 activations, split-bf16 (hi*hi + hi*lo + lo*hi) MFMA, fp32
 *                           accumulate (~2^-16 per product: meets 1e-3 relative on logits)
 *       CRIMAC_PREC_F32X6 : fp32 activations, 3-plane split, 6 MFMAs per product (~2^-24 per product,
 *                           i.e. fp32-equivalent: the parity mode for bit-exact argmax masks)
 *       CRIMAC_PREC_FP16  : fp16 activations, fp16 MFMA (v_mfma_f32_16x16x32_f16), fp32 accumulate; same
 *                           kernels and rate as BF16, 3 more mantissa bits, 5-bit exponent -> the caller
 *                           scales the loss gradient (loss scaling)
 *     weight operands: `w_hi` is plane 0; `w_lo` holds the remaining (planes-1) planes back to back
 *     (unused for BF16)
 *     parameters, gradients of parameters, statistics and the loss are always fp32 / fp64
 */
#ifndef CRIMAC_UNET_HIP_H_
#define CRIMAC_UNET_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define CRIMAC_OK 0
#define CRIMAC_ERR_INVALID (-22) /* bad argument (EINVAL) */
#define CRIMAC_ERR_LAUNCH (-5)   /* kernel launch failed (EIO) */

#define CRIMAC_PREC_BF16 0
#define CRIMAC_PREC_F32X3 1
#define CRIMAC_PREC_F32X6 2
#define CRIMAC_PREC_FP16 3  /* fp16 activations / gradients, fp16 MFMA, fp32 accumulate (BASELINE configs[4]);
                             * gradients need loss scaling: crimac_wce_bwd's `upstream`, undone by
                             * crimac_sgd_momentum's `grad_scale`, overflow guard crimac_grad_overflow_flag */
#define CRIMAC_PREC_F32H3 4 /* fp32 activations, FORWARD products on a 2-plane fp16 split (hi*hi + hi*lo + lo*hi, 11 + 11
                             * mantissa bits: ~2^-21 per product, fp32-class) at 3 MFMAs per product -- half the MFMAs
                             * of F32X6.  fp16 planes only have the range for forward operands (activations O(1..100),
                             * weights pre-scaled by 2^CRIMAC_F32H3_WSHIFT when packed, undone in the conv epilogue);
                             * gradients span too many decades, so the backward kernels of this mode are called with
                             * CRIMAC_PREC_F32X3 (entry points that only exist for the backward pass reject F32H3). */
#define CRIMAC_PREC_H3P 5   /* PRE-SPLIT fp16 plane pairs: the arithmetic of F32H3 (hi*hi + hi*lo + lo*hi on fp16 planes, 3 MFMAs per
                             * product, fp32 accumulate, ~2^-21 per product) with the split done ONCE by the kernel that PRODUCES
                             * a tensor instead of by every consumer while it stages its tiles.  MFMA operand tensors
                             * (activations, the gradients dy that feed input- and weight-gradient contractions) are stored as
                             * "plane pairs": addressed like fp32 (4 bytes per element, `ld` in elements) but every aligned
                             * group of 8 channels holds [8 x fp16 hi][8 x fp16 lo], value = hi + lo.  The 16-bit LDS-DMA
                             * kernels then take them as 16-bit tensors of twice the channels.  Tensors only elementwise
                             * kernels read (convolution outputs in training, activation gradients da) stay fp32.  Forward AND
                             * backward run on fp16 planes: gradients travel scaled by the caller's loss scale exactly as in
                             * CRIMAC_PREC_FP16 (crimac_wce_bwd `upstream`, crimac_sgd_momentum `grad_scale`, overflow guard).
                             * Weight planes: crimac_pack_* with CRIMAC_PLANES_H3P (interleaved [32 hi | 32 lo] per 32-channel
                             * block, both directions fp16 and pre-scaled by 2^CRIMAC_F32H3_WSHIFT; `w_lo` is unused). */
#define CRIMAC_PREC_MAX 5
/* Backward-pass precision of the engine's 'h3f' mode (forward pass = CRIMAC_PREC_H3P): the MFMA operands of the backward
 * pass are plain fp16 (1 MFMA per product) -- output gradients dy [pixels][ld] of 2-byte halves, input-gradient weight
 * planes packed with CRIMAC_PLANES_FP16 -- while everything else keeps its H3P format: conv outputs y and activation
 * gradients da fp32 (so ReLU masks, xhat, max-pool positions are decided on exactly the forward pass's values), forward
 * activations fp16 plane pairs (a 16-bit contraction reads their hi plane: the fp16 rounding of the value, the first 16
 * bytes of each 32-byte group).  Accepted ONLY by: crimac_conv3x3 / crimac_conv3x3_cols (input fp16, output fp32, or fp16
 * with CRIMAC_EPI_OUT_PLANES), crimac_upconv2x2_dgrad_bnb_prec (input fp16, output fp32), crimac_wgrad / crimac_wgrad_group
 * (the GRADIENT operand fp16 -- F for mode 0, S for mode 1 --, the ACTIVATION operand plane pairs addressed in 4-byte
 * elements), crimac_bn_bwd_apply / crimac_bn_bwd_apply_replicas (da, y fp32 -> dy fp16).  The reference has no
 * counterpart (fp32 everywhere). */
#define CRIMAC_PREC_H3F_BWD 6
#define CRIMAC_F32H3_WSHIFT 8
/* `planes` argument of the weight-packing entry points: 1..3 bf16 planes (BF16 / F32X3 / F32X6), or
 * CRIMAC_PLANES_FP16 = one IEEE-half plane each way (FP16), or CRIMAC_PLANES_F32H3.  Layout of the argument: bits
 * 0-3 number of planes, bit 4 / bit 5 forward / input-gradient planes in IEEE half, bits 8-15 log2 of the scale
 * applied to the forward planes. */
#define CRIMAC_PLANES_FWD_FP16 16   /* forward planes are IEEE half instead of bf16 */
#define CRIMAC_PLANES_DG_FP16 32    /* input-gradient planes are IEEE half instead of bf16 */
#define CRIMAC_PLANES_FP16 (1 | CRIMAC_PLANES_FWD_FP16 | CRIMAC_PLANES_DG_FP16)                /* FP16 storage mode */
/* F32H3: two fp16 forward planes of w * 2^CRIMAC_F32H3_WSHIFT (bits 8-15 carry the shift), two bf16 dgrad planes */
#define CRIMAC_PLANES_F32H3 (2 | CRIMAC_PLANES_FWD_FP16 | (CRIMAC_F32H3_WSHIFT << 8))
#define CRIMAC_PLANES_INTERLEAVED 64 /* both planes in ONE buffer (`*_hi`; `*_lo` unused), per row of K input channels:
                                      * [CB hi | CB lo] per block of CB = 32 channels (CB = K when K < 32) */
#define CRIMAC_PLANES_FWD_FRAG 65536 /* crimac_pack_conv3x3 only: the FORWARD plane is written fragment-major (CRIMAC_EPI_WFRAG);
                                     * single 16-bit plane: Co % 32 == 0, Ci_pad % 64 == 0; interleaved pairs: Ci_pad % 32 == 0 */
#define CRIMAC_PLANES_DG_SCALED 128  /* the scale of bits 8-15 is applied to the input-gradient planes as well */
/* H3P: interleaved fp16 plane pairs, forward and input-gradient planes, both pre-scaled by 2^CRIMAC_F32H3_WSHIFT */
#define CRIMAC_PLANES_H3P (2 | CRIMAC_PLANES_FWD_FP16 | CRIMAC_PLANES_DG_FP16 | CRIMAC_PLANES_INTERLEAVED | \
                           CRIMAC_PLANES_DG_SCALED | (CRIMAC_F32H3_WSHIFT << 8))

/* `relu` argument of the convolution entry points: bit 0 = ReLU on the output; bit 1 (CRIMAC_PREC_H3P only, no fused
 * reduction) = store the output as fp16 plane pairs (it feeds another contraction) instead of fp32. */
#define CRIMAC_EPI_RELU 1
#define CRIMAC_EPI_OUT_PLANES 2
/* bit 2 (CRIMAC_PREC_H3P, Cin == 16): a promise that only the first FOUR input channels are non-zero (the network input,
 * unet.py:77 with in_channels <= 4) -- selects the persistent first-layer kernel, which contracts their hi / lo planes as
 * sixteen pseudo-channels.  Without it the generic kernel runs. */
#define CRIMAC_EPI_CIN4 4
/* bit 3: with stat_mode 1 the sums are taken of the fp32 results BEFORE the rounding of the store (they are then a bias
 * gradient -- the transposed convolution's, unet.py:130 -- not the statistics of the stored tensor) */
#define CRIMAC_EPI_STAT_RAW 8
/* bit 4: the weight plane w_hi is FRAGMENT-MAJOR (packed by crimac_pack_layers for a layer whose `kind` carries
 * CRIMAC_LAYER_FWD_FRAG / CRIMAC_LAYER_DG_FRAG): element (tap t, row r of R = N, column c of K = Cin) at
 *   ((((t * R/32 + r/32) * K/64 + c/64) * 4 + ((c % 64) / 32) * 2 + (r % 32) / 16) * 64 + ((c % 32) / 8) * 16 + r % 16) * 8 + c % 8
 * -- the four MFMA weight fragments of a (tap, 32-row block, 64-column chunk) are four consecutive kilobytes with lane l's
 * 16 bytes at 16 l, so that every weight load of the channel-split kernel reads whole cache lines.  16-bit precisions
 * (BF16 / FP16), N and the channel range multiples of 128, Cin % 64 == 0.  Plane pairs (H3P; interleaved rows of 2 Cin halves
 * [32 hi | 32 lo] per 32-channel block): the same formula on the row of halves, K = 2 Cin, c = the half's position in the row
 * (Cin % 32 == 0).  H3F_BWD: its fp16 input-gradient planes, as FP16. */
#define CRIMAC_EPI_WFRAG 16
/* ... and (with CRIMAC_EPI_WFRAG) the convolution runs the ROWS form of the channel-split kernel: 64-channel x (32 x 16)-pixel
 * tiles, a wave = 16 channels x 512 pixels, half the weight bytes per MFMA.  16-bit precisions and H3F_BWD; N and the channel
 * range multiples of 64, Cin % 64 == 0.  Same results up to the order of the fp32 accumulation. */
#define CRIMAC_EPI_WROWS 32

/* Library identity / error text.  crimac_version() returns CRIMAC_ABI_VERSION of the build: it is bumped whenever a
 * struct passed by pointer (crimac_layer_desc), the meaning of an argument or the set of precisions changes, and a
 * binding must refuse a library whose version differs from the header it was written against (an older build that
 * happens to export every symbol would walk a descriptor array with the wrong stride).  crimac_layer_desc_size() is
 * sizeof(crimac_layer_desc) as the library was compiled. */
#define CRIMAC_ABI_VERSION 6
int crimac_version(void);
int crimac_layer_desc_size(void);
const char* crimac_last_error(void);

/* ---- dense contractions (MFMA) ------------------------------------------------------------- */

/* Implicit-GEMM convolution: nn.Conv2d 3x3 pad 1 forward (unet.py:35-44, used at :77,:80,:114-119),
 * its input gradient (aten::convolution_backward, pipeline.py:177), nn.ConvTranspose2d k2 s2
 * forward (unet.py:47-49, :130) and its input gradient.
 *   rows m=(b,oy,ox) on [B][Ho][Wo]; tap t reads input pixel (oy*stride + t/tw - pad,
 *   ox*stride + t%tw - pad) of [B][Hi][Wi] (zero outside); w_hi/w_lo: bf16 [ntaps][N][Cin]
 *   (w_lo only for F32X3); bias fp32 indexed n % bias_mod (NULL = none); relu != 0 clamps at 0;
 *   out_mode 0: out[m*out_ld + n]; out_mode 1 (transposed conv): n=(a*2+b)*cout_up+o is scattered
 *   to pixel (2oy+a, 2ox+b) of [B][2Ho][2Wo], channel o.  Cin % 16 == 0, N % 64 == 0. */
int crimac_igemm_conv(int prec, const void* in, long in_ld, int B, int Hi, int Wi, int Ho, int Wo,
                      int Cin, int N, int ntaps, int tw, int pad, int stride, const void* w_hi,
                      const void* w_lo, const float* bias, int bias_mod, void* out, long out_ld,
                      int relu, int out_mode, int cout_up, void* stream);

/* Input gradient of nn.ConvTranspose2d k2 s2 (unet.py:47-49, :130; aten::convolution_backward, pipeline.py:177)
 * when the result is the gradient `da` of a BatchNorm+ReLU block's output (unet.py:121-122, :135-136): that
 * block's BatchNorm-backward sums are accumulated in the epilogue exactly as crimac_conv3x3 stat_mode 2 does
 * (bnb_y = the block's saved conv output on the COARSE grid [B][H][W], bnb_vec rows mean | invstd | scale |
 * shift with row stride bnb_stride, fp64 accumulators [stat_replicas][Cin]).  bf16 only.
 *   dy [B][2H][2W] (dy_ld) fine grid, Cout channels; w_dg_hi: dgrad planes of crimac_pack_upconv2x2 /
 *   crimac_pack_layers, bf16 [4][Cin][Cout]; dx [B][H][W] (dx_ld), Cin channels.
 *   Cout % 64 == 0, Cin % 128 == 0, dy below 2 GiB; otherwise an error (use crimac_igemm_conv +
 *   crimac_bn_bwd_reduce). */
int crimac_upconv2x2_dgrad_bnb(const void* dy, long dy_ld, int B, int H, int W, int Cout, int Cin,
                               const void* w_dg_hi, void* dx, long dx_ld, const void* bnb_y, long bnb_y_ld,
                               const float* bnb_vec, long bnb_stride, double* stat_sum, double* stat_sumsq,
                               int stat_replicas, void* stream);
/* The same for either 16-bit storage mode (prec = CRIMAC_PREC_BF16 or CRIMAC_PREC_FP16). */
int crimac_upconv2x2_dgrad_bnb_prec(int prec, const void* dy, long dy_ld, int B, int H, int W, int Cout, int Cin,
                                    const void* w_dg_hi, void* dx, long dx_ld, const void* bnb_y, long bnb_y_ld,
                                    const float* bnb_vec, long bnb_stride, double* stat_sum, double* stat_sumsq,
                                    int stat_replicas, void* stream);


/* 3x3 convolution, stride 1, pad 1 (nn.Conv2d forward, unet.py:35-44; with the dgrad weight planes
 * of crimac_pack_conv3x3 it is the input gradient): halo tile staged once per channel chunk in LDS,
 * weights streamed two steps ahead, coalesced epilogue.  w_hi/w_lo: bf16 [9][N][Cin].
 * Fused per-channel reductions into fp64 accumulators stat_sum / stat_sumsq [stat_replicas][N] (caller
 * zeroes; workgroups spread over the replicas to avoid same-address atomic serialisation):
 *   stat_mode 0: none;
 *   stat_mode 1: sum / sum of squares of the STORED output -- BatchNorm batch statistics fused into the
 *                convolution that feeds it (unet.py:78,81,121-122; finish with crimac_bn_finalize);
 *   stat_mode 2: the output is `da` of a BatchNorm+ReLU backward: with y = bnb_y [pixels][bnb_y_ld] (that
 *                layer's saved conv output) and bnb_vec rows (mean, invstd, scale, shift; row stride
 *                bnb_stride): stat_sum += sum dz, stat_sumsq += sum dz*xhat, dz = da*[y*scale+shift > 0]
 *                -- crimac_bn_bwd_reduce fused into the input-gradient convolution (finish with
 *                crimac_sum_replicas, then crimac_bn_bwd_apply).
 * Cin % 16 == 0, N % 64 == 0. */
int crimac_conv3x3(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                   const void* w_hi, const void* w_lo, const float* bias, void* out, long out_ld,
                   int relu, int stat_mode, double* stat_sum, double* stat_sumsq, int stat_replicas,
                   const void* bnb_y, long bnb_y_ld, const float* bnb_vec, long bnb_stride, void* stream);

/* crimac_conv3x3 restricted to the output channels [n_first, n_first + n_count) of the N-channel convolution;
 * every pointer (weights, bias, out, statistic accumulators) is that of the FULL convolution.  Used for the
 * input gradient of a decoder block's first convolution (unet.py:114-119 after the concat at :132): the half
 * that feeds the transposed convolution's backward is produced at once, the skip-connection half
 * (encoder_outs, unet.py:336) on a side stream -- it is not needed until the encoder level is reached.
 * bf16, Cin % 64 == 0; ranges in multiples of 128 channels, or 64 channels of a 64-input-channel convolution
 * with >= 512 tiles of 16x16 pixels; anything else is an error. */
int crimac_conv3x3_cols(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                   const void* w_hi, const void* w_lo, const float* bias, void* out, long out_ld,
                   int relu, int stat_mode, double* stat_sum, double* stat_sumsq, int stat_replicas,
                   const void* bnb_y, long bnb_y_ld, const float* bnb_vec, long bnb_stride, int n_first, int n_count, void* stream);

/* Eval-mode encoder block tail: crimac_conv3x3 (bias / folded BatchNorm / ReLU) that ALSO writes
 * pool_out [B][H/2][W/2][pool_ld] = nn.MaxPool2d(2, 2) (unet.py:85-86) of its stored output from the epilogue's
 * staged tile -- no second pass over the skip tensor.  H, W even; not for the first-layer shape (Cin 16 -> 64). */
int crimac_conv3x3_pool(int prec, const void* in, long in_ld, int B, int H, int W, int Cin, int N,
                        const void* w_hi, const void* w_lo, const float* bias, void* out, long out_ld, int relu,
                        void* pool_out, long pool_ld, void* stream);

/* Weight gradient (aten::convolution_backward weight half, pipeline.py:177):
 *   dw[t][f][s] += sum_pixels F[p][f] * S[shift_t(p)][s]   (fp32 atomics; caller zeroes dw)
 *   mode 0 (conv3x3): F=dY [B][Hf][Wf][CF=Cout], S=X same grid [CS=Cin], 9 taps
 *   mode 1 (upconv):  F=X  [B][Hf][Wf][CF=Cin],  S=dY [B][2Hf][2Wf][CS=Cout], 4 taps (a,b)
 *   target_blocks: workgroups to aim for when splitting the pixel range; <= 0: automatic (about
 *   512 workgroups = one resident round, but never fewer than ~4096 pixels per split: each split costs
 *   one atomic pass). */
int crimac_wgrad(int prec, int mode, const void* f, long f_ld, int CF, const void* s, long s_ld,
                 int CS, int B, int Hf, int Wf, float* dw, int target_blocks, void* stream);

/* The conv3x3 weight gradients of SEVERAL layers in one persistent launch (16-bit storage precisions and plane pairs,
 * Cin >= 64; plane pairs: whole 64-channel tiles): the
 * same contraction and the same fp32-atomic accumulation into each layer's dw as crimac_wgrad(mode 0), but the (layer,
 * 64 x 64 channel tile, pixel split) items of all layers go through per-XCD queues that persistent workgroups pull
 * from, so the atomic flush of a finished item drains under the MFMAs of the next one instead of ending every launch
 * with all workgroups flushing at once (~30 us x 22 launches per step).  The reference computes these gradients inside
 * aten::convolution_backward, one layer at a time (loss.backward(), pipeline.py:177).
 *   crimac_wgrad_group_plan (HOST ONLY, no stream): fills the planned fields of layers[] and the queues -- items
 *     [8][cap][2] int32 host array (NULL: sizing call), counts[8]; returns the queue capacity the plan needs (>= 0) or a
 *     negative error.  items_per_layer <= 0: default (128, CRIMAC_WGRAD_GROUP_ITEMS overrides).  The plan depends on the
 *     shapes and B only, not on the pointers.
 *   crimac_wgrad_group: layers[] as planned (f, s, dw set), items = the DEVICE copy of the planned queues, counters =
 *     8 device uint32 ZEROED by the caller before every launch. */
#define CRIMAC_WGRAD_GROUP_MAX_LAYERS 16
typedef struct crimac_wgrad_group_layer {
  const void* f; long f_ld; int CF;      /* dY  [B][Hf][Wf][CF = Cout] */
  const void* s; long s_ld; int CS;      /* X   [B][Hf][Wf][CS = Cin]  */
  int Hf, Wf;
  float* dw;                             /* [9][CF][CS] fp32, accumulated into (caller zeroes) */
  int tiles_y, tiles_x;                  /* planned fields */
  long ntiles;
  int tiles_per_block, nsplits;
} crimac_wgrad_group_layer;
int crimac_wgrad_group_layer_size(void);   /* sizeof(crimac_wgrad_group_layer) as the library was compiled */
int crimac_wgrad_group_plan(int prec, crimac_wgrad_group_layer* layers, int n_layers, int B, int items_per_layer,
                            int* items, int cap, int* counts);
int crimac_wgrad_group(int prec, const crimac_wgrad_group_layer* layers, int n_layers, int B, const int* items, int cap,
                       const int* counts, unsigned int* counters, void* stream);
/* The same contraction WITHOUT atomics: the pixel range is split into crimac_wgrad_splits(...) parts and the
 * workgroups of part k store their finished 64 x 64 x taps tiles into slab k = partials + k * slab_stride (plain
 * stores; every element of every slab is written, no zero fill needed).  crimac_unpack_wgrad_layers adds the slabs
 * up in order, so the weight gradient is bit-reproducible run to run; it also removes the fp32-atomic tail of the
 * single resident round (each split was one atomic pass over dW at ~1.3 TB/s chip-wide).  slab_stride in floats,
 * >= taps * CF * CS.  crimac_wgrad_splits returns the number of slabs for a shape (> 0) or a negative error. */
int crimac_wgrad_splits(int prec, int mode, int CF, int CS, int B, int Hf, int Wf, int target_blocks);
int crimac_wgrad_partials(int prec, int mode, const void* f, long f_ld, int CF, const void* s, long s_ld, int CS,
                          int B, int Hf, int Wf, float* partials, long slab_stride, int target_blocks, void* stream);


/* ---- weight layout (fp32 master weights <-> bf16 MFMA operand planes) ------------------------ */

/* Conv2d weight [Co][Ci][3][3] -> fwd planes [9][Co][Ci_pad] (scaled per Co by `scale` if given:
 * eval-mode BatchNorm folding) and dgrad planes [9][Ci][Co] (flipped taps).
 * `planes` (1..3): number of bf16 planes the value is split into; *_lo receive planes 1..planes-1
 * back to back (may be NULL when planes == 1); dgrad planes may be NULL. */
int crimac_pack_conv3x3(const float* w, int Co, int Ci, int Ci_pad, const float* scale, int planes,
                        void* fwd_hi, void* fwd_lo, void* dg_hi, void* dg_lo, void* stream);
/* ConvTranspose2d weight [Ci][Co][2][2] -> fwd planes [(a,b,o)][Ci], dgrad planes [(a,b)][Ci][Co]. */
int crimac_pack_upconv2x2(const float* w, int Ci, int Co, int planes, void* fwd_hi, void* fwd_lo,
                          void* dg_hi, void* dg_lo, void* stream);
/* dw [9][Co][Ci_pad] -> grad [Co][Ci][3][3];  dw [4][Ci][Co] -> grad [Ci][Co][2][2]. */
int crimac_unpack_wgrad_conv3x3(const float* dw, int Co, int Ci, int Ci_pad, float* grad, void* stream);
int crimac_unpack_wgrad_upconv2x2(const float* dw, int Ci, int Co, float* grad, void* stream);

/* The same re-layouts for EVERY layer of the network in one launch each way (the weights change every
 * step, optim.SGD.step pipeline.py:178, so the planes are rebuilt every step).  `descs` is a HOST array;
 * it is read during the call only.  Co % 32 == 0; kind 1 also needs Ci % 32 == 0. */
typedef struct crimac_layer_desc {
  const float* w;      /* fp32 weights: kind 0 [Co][Ci][3][3], kind 1 [Ci][Co][2][2] (device) */
  float* grad;         /* crimac_unpack_wgrad_layers: gradient, laid out as w (device) */
  const float* dw;     /* crimac_unpack_wgrad_layers: packed gradient written by crimac_wgrad (device) */
  void* fwd_hi;        /* crimac_pack_layers: planes as in crimac_pack_conv3x3 / crimac_pack_upconv2x2 */
  void* fwd_lo;
  void* dg_hi;         /* may be NULL (no input gradient needed: first layer) */
  void* dg_lo;
  int kind;            /* bit 0 -- 0: Conv2d 3x3, 1: ConvTranspose2d 2x2 stride 2; kind 0, 16-bit single-plane or interleaved-pair packs:
                        * | CRIMAC_LAYER_FWD_FRAG: fwd_hi is written FRAGMENT-MAJOR (CRIMAC_EPI_WFRAG; Co % 32 == 0, Ci_pad % 64 == 0),
                        * | CRIMAC_LAYER_DG_FRAG: dg_hi likewise (rows = Ci, columns = Co: Ci % 32 == 0, Co % 64 == 0) */
  int Co, Ci, Ci_pad;  /* Ci_pad: kind 0 only */
  int dw_splits;       /* crimac_unpack_wgrad_layers: `dw` holds this many partial slabs (crimac_wgrad_partials),
                        * added up in a fixed order (bit-reproducible; more than 16 slabs are first folded IN PLACE
                        * into the first 16, so dw is scratch afterwards); 0 or 1: dw is the finished sum (crimac_wgrad) */
  long dw_stride;      /* floats between two slabs */
} crimac_layer_desc;
#define CRIMAC_LAYER_FWD_FRAG 16
#define CRIMAC_LAYER_DG_FRAG 32
int crimac_pack_layers(const crimac_layer_desc* descs, int n_layers, int planes, void* stream);
int crimac_unpack_wgrad_layers(const crimac_layer_desc* descs, int n_layers, void* stream);

/* ---- layout ---------------------------------------------------------------------------------- */

/* Model input [B][C][H][W] fp32 (pipeline.py:163, :208 `.float().to(device)`) -> NHWC activations
 * with channels zero-padded to ld. */
int crimac_nchw_to_nhwc(int prec, const float* in, void* out, int B, int C, int H, int W, long ld,
                        void* stream);

/* ---- BatchNorm2d / ReLU / MaxPool2d (unet.py:78-86, :121-122, :135-136) ---------------------- */

/* Per-channel sum (and sum of squares if sumsq != NULL) over M pixels, accumulated into fp64. */
int crimac_colstats(int prec, const void* y, long ld, long M, int C, double* sum, double* sumsq,
                    void* stream);
/* Per-channel sum accumulated into fp32 (bias gradients). */
int crimac_colsum_f32(int prec, const void* y, long ld, long M, int C, float* sum, void* stream);
/* dst[i] = sum_r src[r*stride + i], i < n, into fp64 and/or fp32 (either may be NULL); if src_b is
 * given, dst_b_f64[i] = sum_r src_b[r*stride + i] in the same launch (the two BatchNorm-backward sums). */
int crimac_sum_replicas(const double* src, int replicas, long stride, int n, double* dst_f64,
                        float* dst_f32, const double* src_b, double* dst_b_f64, void* stream);
/* Train-mode statistics -> mean, invstd, scale=gamma*invstd, shift=beta-mean*scale; running stats
 * updated with `momentum` (unbiased variance), num_batches_tracked += 1.
 * sum/sumsq are [replicas][C] partial accumulators that are added up first. */
int crimac_bn_finalize(const double* sum, const double* sumsq, int replicas, long M, int C, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, long long* num_batches_tracked, float* mean,
                       float* invstd, float* scale, float* shift, void* stream);
/* out = [relu](y*scale+shift) (scale NULL = identity); optional 2x2/2 max-pool of the result.
 * out and pool_out are each optional. */
int crimac_bn_act_pool(int prec, const void* y, long y_ld, const float* scale, const float* shift,
                       int relu, void* out, long out_ld, void* pool_out, long pool_ld, int B, int H,
                       int W, int C, void* stream);
/* crimac_bn_finalize + crimac_bn_act_pool in ONE launch (train mode: nn.BatchNorm2d + ReLU [+ MaxPool2d],
 * unet.py:78-86, :121-122): every workgroup adds up the convolution epilogue's [replicas][C] accumulators itself
 * (count = pixels they cover) and applies scale / shift; workgroup 0 writes the rows mean | invstd | scale | shift of
 * bn_vec (row stride bn_stride; the backward pass reads them) and updates the running statistics as
 * crimac_bn_finalize does.  Keep replicas * C small (<= 4096: one batch of loads per thread): that many fp64 pairs are read per workgroup. */
int crimac_bn_train_act_pool(int prec, const void* y, long y_ld, const double* stat_sum, const double* stat_sumsq,
                             int replicas, long count, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float* bn_vec, long bn_stride, int relu, void* out,
                             long out_ld, void* pool_out, long pool_ld, int B, int H, int W, int C, void* stream);
/* Backward of [pool ->] (skip add): da = ds + unpool(dp) with first-max tie rule of
 * aten::max_pool2d; `a` is the forward activation that was pooled.  ds may be NULL.
 * stat_sum != NULL: `da` feeds a BatchNorm+ReLU block; its backward sums (as crimac_conv3x3 stat_mode 2:
 * y = bnb_y, rows of bnb_vec = mean, invstd, scale, shift) are accumulated into [stat_replicas][C] while
 * da is produced (crimac_bn_bwd_reduce fused away; finish with crimac_sum_replicas).  In that form `a` MUST be
 * that block's stored activation round(relu(bnb_y * scale + shift)): the kernel rebuilds it from bnb_y (which it
 * reads for the sums anyway) instead of reading `a` -- bit-identical, one tensor less.  crimac_head_bwd
 * takes the same seven arguments for its dx. */
int crimac_unpool_add(int prec, const void* dp, long dp_ld, const void* a, long a_ld, const void* ds,
                      long ds_ld, void* da, long da_ld, int B, int H, int W, int C, const void* bnb_y,
                      long bnb_y_ld, const float* bnb_vec, long bnb_stride, double* stat_sum,
                      double* stat_sumsq, int stat_replicas, void* stream);
/* BatchNorm+ReLU backward, pass 1: sum_dz, sum_dz_xhat (fp64, caller zeroes) with
 * dz = da * (y*scale+shift > 0), xhat = (y-mean)*invstd. */
int crimac_bn_bwd_reduce(int prec, const void* da, long da_ld, const void* y, long y_ld,
                         const float* scale, const float* shift, const float* mean,
                         const float* invstd, long M, int C, double* sum_dz, double* sum_dz_xhat,
                         void* stream);
/* pass 2 over M pixels: dy = scale*(dz - sum_dz/count - xhat*sum_dz_xhat/count), count = number of pixels
 * the sums were taken over (0 = M; larger when they were all-reduced over ranks, SyncBN); dgamma =
 * sum_dz_xhat, dbeta = sum_dz;
 * dbias (conv bias in front of the BN, may be NULL) += sum over pixels of dy (caller zeroes). */
int crimac_bn_bwd_apply(int prec, const void* da, long da_ld, const void* y, long y_ld,
                        const float* scale, const float* shift, const float* mean, const float* invstd,
                        const double* sum_dz, const double* sum_dz_xhat, long M, long count, int C, void* dy,
                        long dy_ld, float* dgamma, float* dbeta, float* dbias, void* stream);

/* crimac_sum_replicas + crimac_bn_bwd_apply in ONE launch: sum_dz / sum_dz_xhat are the [replicas][C] accumulators the
 * producer of `da` filled (stat_mode 2 / crimac_unpool_add / crimac_head_bwd); bn_vec rows mean | invstd | scale | shift. */
int crimac_bn_bwd_apply_replicas(int prec, const void* da, long da_ld, const void* y, long y_ld, const float* bn_vec,
                                 long bn_stride, const double* sum_dz, const double* sum_dz_xhat, int replicas, long M,
                                 long count, int C, void* dy, long dy_ld, float* dgamma, float* dbeta, void* stream);

/* Round 4: crimac_unpool_add (called with da == NULL: it then only takes the BatchNorm-backward sums) followed by
 * crimac_bn_bwd_apply_replicas, without the round trip of da = ds + unpool(dp) through HBM: da is rebuilt from (dp, ds, y)
 * with the arithmetic and the storage rounding of crimac_unpool_add while dy is formed.  Backward of
 * MaxPool2d(2,2) + skip + BatchNorm2d + ReLU of one encoder level (reference unet.py:85-92, :78-82).  ds may be NULL. */
int crimac_unpool_bn_bwd_apply_replicas(int prec, const void* dp, long dp_ld, const void* ds, long ds_ld, const void* y,
                                        long y_ld, const float* bn_vec, long bn_stride, const double* sum_dz,
                                        const double* sum_dz_xhat, int replicas, long count, void* dy, long dy_ld, int B,
                                        int H, int W, int C, float* dgamma, float* dbeta, void* stream);

/* ---- 1x1 head, loss, optimiser --------------------------------------------------------------- */

/* conv_final 1x1 (unet.py:59-60, :342): logits [B][ncls][H][W] fp32 (NCHW, as the reference
 * returns them); softmax != 0 applies F.softmax(dim=1) (pipeline.py:218). ncls in 2..4.
 * bn_scale / bn_shift (both or neither; NULL = x is the activation itself): x is the raw conv output of
 * the last BatchNorm block (unet.py:121-122, :136) and its BatchNorm+ReLU is applied on the fly -- the
 * activation is never written.  crimac_head_bwd with x == NULL (only together with the fused sums, bnb_*)
 * rebuilds it the same way from bnb_y and the scale / shift rows of bnb_vec. */
int crimac_head_fwd(int prec, const void* x, long x_ld, int Cin, const float* w, const float* b,
                    float* logits, int B, int H, int W, int ncls, int softmax, const float* bn_scale,
                    const float* bn_shift, void* stream);
int crimac_head_bwd(int prec, const float* dlogits, const void* x, long x_ld, int Cin, const float* w,
                    void* dx, long dx_ld, float* dw, float* db, int B, int H, int W, int ncls,
                    const void* bnb_y, long bnb_y_ld, const float* bnb_vec, long bnb_stride, double* stat_sum,
                    double* stat_sumsq, int stat_replicas, void* stream);
/* F.softmax(outputs, dim=1) on NCHW logits (SegPipe.predict_batch(return_softmax=True), pipeline.py:218; the validation
 * loop, pipeline.py:269, where the logits themselves are still needed for the loss): out[b][c][h][w] fp32.  (When only the
 * probabilities are wanted crimac_head_fwd(softmax = 1) produces them directly.) */
int crimac_softmax_nchw(const float* logits, float* out, int B, int ncls, int H, int W, void* stream);

/* nn.CrossEntropyLoss(weight) (pipeline.py:132-141): sums[0] += sum w[y]*nll, sums[1] += sum w[y]
 * over pixels with y != ignore_index.  labels: int16/int32/int64 selected by label_bytes. */
int crimac_wce_fwd(const float* logits, const void* labels, int label_bytes, const float* class_w,
                   int ncls, int ignore_index, int B, int H, int W, double* sums, void* stream);
/* dlogits = upstream * w[y]/sums[1] * (softmax - onehot); 0 where ignored. */
int crimac_wce_bwd(const float* logits, const void* labels, int label_bytes, const float* class_w,
                   int ncls, int ignore_index, int B, int H, int W, const double* sums,
                   float upstream, float* dlogits, void* stream);
/* optim.SGD(lr, momentum) (pipeline.py:156, :178) over a flat buffer:
 * g' = g*grad_scale; v = momentum*v + g'; p -= lr*v; g = 0 if zero_grad. */
int crimac_sgd_momentum(float* p, float* g, float* v, long n, float lr, float momentum,
                        float grad_scale, int zero_grad, void* stream);
/* Loss-scaled training (CRIMAC_PREC_FP16; the reference trains in fp32 and needs none): state[0] |= 1 if any of
 * the n gradients is inf / NaN (caller zeroes state[0] before every step); crimac_sgd_momentum_guarded then leaves
 * p and v untouched for that step and counts it in state[1].  No host synchronisation is involved: the caller
 * reads state[1] whenever it flushes its loss log and adapts the scale it passes as crimac_wce_bwd's `upstream`
 * (and divides out again through `grad_scale`). */
int crimac_grad_overflow_flag(const float* g, long n, int* state, void* stream);
int crimac_sgd_momentum_guarded(float* p, float* g, float* v, long n, float lr, float momentum,
                                float grad_scale, int zero_grad, int* state, void* stream);


/* ---- tiled whole-survey inference (save_predict.py:160-209) ------------------------------------ */

/* DatasetGriddedReader.get_preload_data_labels + remove_nan_inf + db_with_limits (dataset.py:192-205,
 * remove_nan_inf.py:23-34, db_with_limits.py:20-24): gather P crops of ph x pw pixels around
 * centres[p] = (range idx, ping idx RELATIVE to the chunk slice) from data [C][Wd pings][H range] fp32
 * (the reader's zarr orientation), 0 outside the slice, non-finite -> 0, 10*log10(x+1e-10) clamped
 * to [-75, 0]; written as NHWC activations [P*ph*pw][ld] (channels >= C zero) -- the first conv's input. */
int crimac_gather_patches(int prec, const float* data, int C, int Wd, int H, const int* centres, int P,
                          int ph, int pw, void* out, long ld, void* stream);
/* The memm flavour of the same crop (save_reader_predictions_memm, save_predict.py:222-265; get_crop_memmap,
 * dataset.py:251-287; define_data_transform_test, transforms.py:57-64): border_labels [Wd][H] int16 raw annotation
 * ids over the same extent as `data`; pixels outside it, and pixels whose raw id the test-time label transform
 * maps to "ignore" (negative ids), are set to 0.0 after the dB transform (set_data_border_value). */
int crimac_gather_patches_memm(int prec, const float* data, int C, int Wd, int H, const int* centres, int P, int ph,
                               int pw, void* out, long ld, const short* border_labels, void* stream);

/* fill_out_array (save_predict.py:41-65) for P patches: probs [P][ncls][ph][pw] fp32 softmax;
 * centres[p] = (range idx, GLOBAL ping idx); writes channels SANDEEL(1), OTHER(2) of every valid
 * interior pixel into out [2][H][n_chunk] fp32 (ping = global ping - start_ping).  A pixel is valid
 * unless: it lies in the `overlap` rim (mask_label_overlap.py:41-46); its raw label (labels
 * [n_chunk][H] int16, NULL = all background) is negative (convert_label_indexing.py:24-47 -> -100);
 * it is background below the seabed (mask_label_seabed.py:24-68: seabed_mask [mask_pings][H] uint8
 * starting at global ping mask_ping0, shifted down by seabed_pad inside the patch slice); channel 0 of
 * the chunk data (data0 [data_pings][H], global ping data_ping0) is non-finite there
 * (remove_nan_inf.py:31); or it falls outside the chunk / the range axis. */
int crimac_scatter_patches(const float* probs, int ncls, const int* centres, int P, int ph, int pw,
                           int overlap, int start_ping, int n_chunk, int H, const short* labels,
                           const unsigned char* seabed_mask, int mask_ping0, int mask_pings,
                           const float* data0, int data_ping0, int data_pings, int seabed_pad, float* out,
                           void* stream);
/* The same with the rules of the other callers of fill_out_array made explicit:
 *   seabed: seabed index per ping [seabed_pings] (global ping of entry 0 = seabed_ping0) -- the mask is then
 *           evaluated in the kernel (range index - pad >= seabed[ping]) and no [pings][H] mask has to be built and
 *           uploaded by the host; give seabed_mask OR seabed;
 *   seabed_rule 0: zarr reader (data_reader.py:837-841: the pad shifts the mask down INSIDE the patch's own slice),
 *               1: memm reader Echogram.get_seabed_mask (data_reader.py:407-431: absolute rows >= seabed + pad);
 *   out_f16 != 0: `out` is float16 -- the reference stores the predictions as float16 (save_predict.py:212, :252),
 *           so the rounding happens here and half the bytes travel back to the host. */
int crimac_scatter_patches_ex(const float* probs, int ncls, const int* centres, int P, int ph, int pw, int overlap,
                              int start_ping, int n_chunk, int H, const short* labels,
                              const unsigned char* seabed_mask, int mask_ping0, int mask_pings, const int* seabed,
                              int seabed_ping0, int seabed_pings, const float* data0, int data_ping0, int data_pings,
                              int seabed_pad, int seabed_rule, void* out, int out_f16, void* stream);


/* Validation metrics (get_predictions_dataloader + compute_evaluation_metrics, pipeline.py:242-295):
 * histograms (16384 bins, indexed by the float16 bit pattern of softmax(logits)[SANDEEL]) of the valid
 * pixels with raw label == SANDEEL (hist_pos) and the others (hist_neg); labels are RAW batch labels:
 * {-70,-30,-100,-10} are skipped, -50 (below seabed) counts as background with probability 0
 * (pipeline.py:222-239, :317).  Accumulates (caller zeroes).  A probability that is not in [0, 1] (NaN logits
 * of a diverged network) is counted in bin CRIMAC_PR_NAN_BIN, which no valid probability reaches (1.0 = 0x3C00);
 * sklearn's precision_recall_curve raises on such input and so does the host wrapper. */
#define CRIMAC_PR_BINS 16384
#define CRIMAC_PR_NAN_BIN (CRIMAC_PR_BINS - 1)
int crimac_pr_histogram(const float* logits, int ncls, const void* labels, int label_bytes, int B, int H,
                        int W, unsigned int* hist_pos, unsigned int* hist_neg, void* stream);

/* Metadata planes of P crops (reference batch/dataset.py:288-351: the `meta` half of get_crop_memmap's result, which the
 * Dataset appends to the data channels, :109 / :241): built from the echogram's scalar `portion_of_year_scalar` and its
 * per-ping vectors `portion_of_day_vector`, `time_vector_diff`, `_seabed` (data_reader.py:98-100) instead of per patch
 * in numpy DataLoader workers.  centres [P][2] = (range idx, ping idx); flags: bit 0 portion_year, bit 1 portion_day (two
 * planes: sin, cos of 2 pi t at the centre ping), bit 2 time_diff (per column), bit 3 depth_rel = row / seabed[ping],
 * bit 4 depth_abs_surface = row / H, bit 5 depth_abs_seabed = (seabed[ping] - row) / H, with row = cy - H/2 + y and
 * ping = cx - W/2 + x, vector indices clamped as the reference does (< 0 -> 0, >= n -> last).  float64 arithmetic, one
 * rounding to out [P][Cm][H][W] fp32 (planes in flag order), as the reference's float64 planes after `.float()`. */
int crimac_meta_planes(const int* centres, int P, int H, int W, int flags, double portion_year,
                       const double* portion_day, int n_day, const double* time_diff, int n_td,
                       const long long* seabed, int n_sb, float* out, void* stream);

/* ---- late metadata injection (UNet_LateMetInject, unet.py:346-391; MetaPostProcessing, unet.py:140-166) ---- */

/* m[b*H*W + p] = W3 . relu(W2 . relu(W1 . meta[b,:,p] + b1) + b2) + b3: the per-pixel 3-layer perceptron
 * (Linear(Cm,32) / ReLU / Linear(32,32) / ReLU / Linear(32,1), applied over the channel axis, unet.py:151-166) on
 * meta [B][Cm][H][W] fp32 NCHW; w1 [32][Cm], w2 [32][32], w3 [1][32] as torch.nn.Linear stores them.  Cm <= 8. */
int crimac_meta_mlp_fwd(const float* meta, int Cm, int B, int H, int W, const float* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3, float* m, void* stream);
/* conv_final on torch.cat((x, m), 1) (unet.py:386-388) = crimac_head_fwd on the first 64 weight columns, then
 * logits[b][o][p] += wm[o] * m[b][p] (wm = conv_final.weight[:, 64]); softmax != 0: F.softmax over the classes
 * afterwards (pipeline.py:218). */
int crimac_meta_inject_fwd(const float* m, const float* wm, float* logits, int B, int H, int W, int ncls, int softmax,
                           void* stream);
/* Backward of both: dwm[o] += sum dlogits[o] * m, dm = sum_o dlogits[o] * wm[o] (recomputed per pixel, never
 * stored), then the gradients of the three Linear layers (accumulated: caller zeroes). */
int crimac_meta_bwd(const float* dlogits, const float* meta, int Cm, int B, int H, int W, int ncls, const float* wm,
                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                    const float* b3, float* dwm, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                    void* stream);

/* ---- on-GPU training augmentation + data transform (BASELINE configs[4]) -------------------------- */

/* add_noise + flip_x_axis (batch/data_augmentation/add_noise.py:21-41, flip_x_axis.py:21-25) fused
 * with remove_nan_inf + db_with_limits and the NCHW->NHWC conversion: data [B][C][H][W] fp32 LINEAR sv,
 * labels_in [B][H][W] int16/32/64 (or NULL) -> out NHWC [B*H*W][ld] dB activations, labels_out int16
 * (NULL to skip; flipped with the data, -100 where channel 0 is non-finite).  With aux_mask [B][H][W]
 * given, labels_out keep the raw ids (no -100 rule) and aux_mask receives bit 0 = thr_lo < augmented linear
 * data[thr_channel] < thr_hi, bit 1 = channel 0 non-finite -- the inputs of crimac_refine_labels, which
 * then runs the reference's label transform in its place (batch/dataset.py:89-103).  Per sample with p=.5:
 * 5 % of the values x U(1,10) or x U(0,1) (half each); per sample with p=.5: ping axis flipped.
 * Randomness: Philox4x32-10 keyed on (seed, sample index), counter = element index.
 * db_scaled != 0: db_with_limits_scaled (db_with_limits.py:27-33: 1 + dB / 75 in [0, 1]), the data transform of the
 * metadata configurations (define_data_transform(use_metadata=True), batch/transforms.py:50-51). */
int crimac_augment_db_nhwc(int prec, const float* data, const void* labels_in, int label_bytes, void* out,
                           short* labels_out, unsigned char* aux_mask, int thr_channel, float thr_lo,
                           float thr_hi, int B, int C, int H, int W, long ld, unsigned long long seed,
                           int do_noise, int do_flip, int db_scaled, void* stream);

/* flip_x_axis_metadata (flip_x_axis.py:27-32) for planes that do not pass through crimac_augment_db_nhwc -- the
 * metadata planes of UNet_LateMetInject (add_noise_metadata, add_noise.py:42-63, leaves them untouched): in [B][C][H][W]
 * fp32 -> out (another buffer), the ping axis of sample b flipped under the same per-sample draw (same seed) as its
 * data and labels. */
int crimac_augment_flip_planes(const float* in, float* out, int B, int C, int H, int W, unsigned long long seed,
                               int do_flip, void* stream);

/* ---- training label transform ------------------------------------------------ */

/* define_label_transform_train (batch/transforms.py:71-78): refine_label_boundary
 * (batch/label_transforms/refine_label_boundary.py:35-104: threshold + 7x7-disk binary closing on the crop
 * of non-boundary labels, unclosed school pixels -> -30), then with mode 1 convert_label_indexing
 * (convert_label_indexing.py:24-35: 0/27/1 -> 0/1/2, rest -> -100) and remove_nan_inf's label rule
 * (remove_nan_inf.py:30-32).  labels_in [B][H][W] raw annotation ids (int16/32/64) -> labels_out int16.
 * The per-pixel data facts come either from `data` [B][C][H][W] fp32 linear sv (thr_lo < data[thr_channel]
 * < thr_hi; channel 0 finite) or, after crimac_augment_db_nhwc, from its aux_mask [B][H][W] (bit 0 / bit 1);
 * pass exactly one of the two.  W <= 1024. */
int crimac_refine_labels(const void* labels_in, int label_bytes, const unsigned char* aux_mask,
                         const float* data, int thr_channel, float thr_lo, float thr_hi, int mode,
                         short* labels_out, int B, int C, int H, int W, void* stream);

/* define_label_transform_test (batch/transforms.py:81-99, label_masks = 'all') + remove_nan_inf's label rule
 * (remove_nan_inf.py:30-32) on a batch of RAW crops -- the label chain of the validation / evaluate flows
 * (pipeline.py:242-341; evaluate.py:39-117): convert_label_indexing_unused_species (convert_label_indexing.py:37-47),
 * refine_label_boundary on the converted labels and the raw linear-sv crop (refine_label_boundary.py:35-104),
 * mask_label_seabed (mask_label_seabed.py:24-68: background below seabed + pad -> -50), mask_label_overlap
 * (mask_label_overlap.py:23-48: rim of `overlap` pixels -> -70 except where the crop left the data).
 *   labels_in [B][H][W] raw annotation ids (-100 outside the data), data [B][C][H][W] fp32 linear sv,
 *   centres [B][2] int64 (range idx, GLOBAL ping idx) -- the batch's `center_coordinates`; the seabed as in
 *   crimac_scatter_patches_ex: per-ping vector seabed[seabed_pings] from global ping seabed_ping0, OR the reader's mask
 *   [mask_pings][n_range] from global ping mask_ping0; seabed_rule 0 = zarr reader (pad applied inside the requested
 *   slice, data_reader.py:837-841), 1 = Echogram (absolute rows, data_reader.py:407-431); H, W even, W <= 1024.
 *   labels_out int16 [B][H][W] in {-100, -70, -50, -30, -10, 0, 1, 2}.  Bit-exact against the reference's output
 *   (tests/golden/labels_test.npz). */
int crimac_labels_test_transform(const void* labels_in, int label_bytes, const float* data, int thr_channel,
                                 float thr_lo, float thr_hi, const long long* centres, const int* seabed,
                                 int seabed_ping0, int seabed_pings, const unsigned char* seabed_mask, int mask_ping0,
                                 int mask_pings, int n_range, int seabed_pad, int seabed_rule, int overlap,
                                 short* labels_out, int B, int C, int H, int W, void* stream);

/* mask_ping0 value of crimac_labels_test_transform: the seabed mask is laid out PER PATCH, [B][W][n_range] (column x of
 * patch b at row b * W + x; columns outside the survey all zero), mask_pings = B * W -- for batches whose patches lie
 * anywhere in a long survey (the reference reads the mask per patch, mask_label_seabed.py:40-52). */
#define CRIMAC_MASK_PER_PATCH (-2147483647 - 1)

/* get_extended_label_mask_for_crop (batch/label_transforms/extend_label_masks.py:35-98): the extra link of
 * define_label_transform_test for eval_mode 'region' / 'trace' (batch/transforms.py:87-90; evaluate.py:50, :92).  In place
 * on the int16 labels crimac_labels_test_transform produced: a pixel keeps its label only inside one of the school
 * bounding boxes, everything else becomes ignore_val (the reference's default: -1); then remove_nan_inf's label rule
 * (remove_nan_inf.py:30-32, applied by the reference after the whole label chain): -100 where data channel 0 is not
 * finite.  boxes [n_boxes][4] int32 = (y0, y1, x0, x1) in echogram coordinates, ALREADY extended by the caller as
 * extend_label_masks.py:70-80 does ('region': all four sides by extend_size; 'trace': y0 = 0, y1 = echogram.shape[0],
 * x by extend_size); the patch is placed at centre - size / 2 as the reference places it (:64).  centres [B][2] int64,
 * data [B][C][H][W] fp32 linear sv; H * W <= 65536. */
int crimac_labels_extend_mask(short* labels, const float* data, int C, const long long* centres, const int* boxes,
                              int n_boxes, int ignore_val, int B, int H, int W, void* stream);

/* ---- measurement support (bench.py only, not on the product path) --------------------------- */

/* MFMA-only calibration launch: `blocks` workgroups of 4 waves each issue iters x 8 v_mfma_f32_16x16x32_bf16 on
 * register operands (no memory traffic): 2 * 16*16*32 * 8 * iters * 4 * blocks FLOP.  stamps (device, 2 x blocks
 * uint64, or NULL) receives per workgroup the s_memtime (shader cycles) and s_memrealtime (100 MHz) differences
 * around the loop: shader clock = stamps[2i] / stamps[2i+1] x 100 MHz.  sink: any device float (never written).
 * The reference has no counterpart (it never measures the device); this exists so that a roofline pass taken on a
 * throttled box is visible in the bench line itself. */
int crimac_mfma_calibrate(int iters, int blocks, unsigned long long* stamps, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRIMAC_UNET_HIP_H_ */
