/*
 * bflow_hip.h -- C ABI of libbflow_hip.so: the MI355X (gfx950 / CDNA4) kernels of the RAFT-spline
 * inference hot path of uzh-rpg/bflow.
 *
 * The reference is 100 % Python/PyTorch and has no native layer; each entry point
 * below replaces the stock-PyTorch op sequence of the cited reference lines (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless stated otherwise, sizes are ints / long long, no torch types;
 *   - the CALLER allocates every buffer; the library never allocates or frees device memory and keeps no
 *     mutable global state (except a thread-local last-error string);
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*): no synchronisation, no
 *     host reads of device data -> safe inside hipGraph stream capture;
 *   - return value 0 = OK, > 0 = hipError_t of the failed launch, < 0 = argument error (BFLOW_E_*);
 *     nothing throws across the ABI.  bflow_last_error_string() describes the last failure on this thread;
 *   - tensors are dense, row-major ("NCHW contiguous") fp32 unless stated otherwise.
 *   - SPLIT FORMAT (every "split" / hi-lo tensor of the matrix-core path): value = hi + lo * 2^-11 with two fp16 planes, ~22
 *     significant bits.  Supported dynamic range of a stored value: |x| <= 65504 -- larger magnitudes SATURATE to +-65504 (never
 *     inf / NaN; NaN inputs stay NaN); 2^-14 <= |x|: 22 bits; |x| < 2^-14: 11 bits (absolute error <= 3e-8).  Products are
 *     accumulated in fp32, so only what is WRITTEN in this format is limited (activations, GRU state, packed weights, look-up output);
 *     pre-normalisation convolution outputs (InstanceNorm statistics, folded BatchNorm) are fp32.  tests/test_hip_parity.py:
 *     test_split_format_*, test_engine_with_large_and_tiny_activations, test_forward_with_rescaled_weights_vs_oracle.
 */
#ifndef BFLOW_HIP_H
#define BFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only this ABI is exported */

#define BFLOW_ABI_VERSION 2   /* 2 (round 6): bflow_voxel_scatter_* became bflow_voxel_grid_* (whole-grid overwrite, workspace argument) and
                               bflow_corr_lookup_conv1x1 was removed in round 5 without a bump; bflow_shader_clock_stamp added.  bflow_amd/hip.py checks it on load */

#define BFLOW_E_ARG      (-1)   /* bad size / null pointer / unsupported configuration */
#define BFLOW_E_LIMIT    (-2)   /* exceeds a compile-time limit (BFLOW_MAX_*) */

#define BFLOW_MAX_PLANES   16   /* (level, target) planes of one correlation pyramid            */
#define BFLOW_MAX_TARGETS   8   /* correlation targets (event targets + image target)           */
#define BFLOW_MAX_DEGREE   16   /* Bezier degree                                                */
#define BFLOW_LOOKUP_RADIUS 4   /* hard-coded in the reference: raft.py:40, corr.py:279         */

typedef void* bflow_stream_t;   /* hipStream_t */

int         bflow_version(void);
const char* bflow_last_error_string(void);
/* Measurement hook (no counterpart in the reference; its CudaTimer, utils/timers.py:11-33, synchronises the device instead): one
 * one-thread launch that writes the device's 100 MHz wall clock into *slot.  Capture-safe, so the stage boundaries of a hipGraph
 * replay can be read back without a tracer (bflow_amd/timers.py StampTimer; bench.py `gpu_stage_ms`).                           */
int         bflow_clock_stamp(unsigned long long* slot, bflow_stream_t stream);
/* Second measurement hook: the SHADER clock under load.  One small launch (2048 one-wave workgroups) that writes, for every CU it reaches,
 * table[row][0] = s_memtime (shader-clock cycles, a counter PER CU) and table[row][1] = s_memrealtime (100 MHz), row = (XCD * 8 + shader
 * engine) * 16 + CU; `table` is BFLOW_CLOCK_TABLE_ROWS x 2 uint64, zeroed by the caller once.  Two stamps (two tables) around a stretch of
 * launches on one stream (or inside one hipGraph) give the average shader clock of that stretch, CU by CU: (d cycles / d ticks) x 100 MHz
 * over the rows both stamps filled.  The part clocks to its power budget (1.3-2.2 GHz under load), so a roofline fraction is only comparable
 * between boxes with this number next to it (bench.py `clock_ghz`).                                                                        */
#define BFLOW_CLOCK_TABLE_ROWS 1024
int         bflow_shader_clock_stamp(unsigned long long* table, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K5  all-pairs correlation volume ("feature_dot_product").
 * Replaces CorrComputation._corr_dot_prod_util, models/raft_utils/corr.py:264-272 (+ the 1-to-N / M-to-N
 * reshapes :237-262):  out[t,b,i,j] = sum_d f1[t?,b,d,i] * f2[t,b,d,j] / sqrt(D).
 *   f1 : (B, D, N) when f1_target_stride == 0 (one reference shared by all T targets), else element
 *        stride between the per-target (B, D, N) blocks (M-to-N; = B*D*N for a dense (T,B,D,N) tensor)
 *   f2 : (T, B, D, N)          out: (T, B, N, N)  == the reference's (T, B*N, 1, h, w) layout
 * fp32 in, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 out.                                        */
int bflow_corr_build_f32(const float* f1, const float* f2, float* out,
                         int T, int B, int D, int N, long long f1_target_stride, bflow_stream_t stream);

/* K5 (fast path)  the same volume on the split-fp16 MFMA engine (csrc/split_gemm.hip): every fp32 operand is carried
 * as hi + lo*2^-11 (two fp16), the dot product as 3 fp16 MFMA chains with fp32 accumulation -> ~2^-22 relative error
 * per product (fp32: 2^-24) at 5.3x the fp32 matrix peak, which makes K5 HBM-write-bound instead of MFMA-bound.
 *   bflow_split_pack      : src (R, D, N) fp32 -> hi, lo (R, Np, D) fp16 (feature-contiguous), rows N..Np zeroed;
 *                           Np = N rounded up to a multiple of 128, D % 8 == 0
 *   bflow_corr_build_split: f1_* (B, Np, D) [f1_target_stride == 0] or per-target blocks f1_target_stride elements
 *                           apart; f2_* (T, B, Np, D); out (T, B, N, N) fp32; D % 32 == 0                        */
/* bflow_corr_build_split_tiled: the same volume with TILED planes, the layout of the inference product path.  The h x w plane of a query
 * pixel is stored as ceil(h/4) x ceil(w/8) tiles of 4 x 8 elements (one fp32 tile = one 128-B line; tiles in row-major order, row-major
 * inside a tile):  element (y, x) at ((y/4)*ceil(w/8) + x/8)*32 + (y%4)*8 + x%8, plane stride = tiles*32.  out (T, B, N, tiles*32) fp32;
 * pad positions of edge tiles hold finite values.  D in {64, 128, 256} only (written by the streaming kernel, csrc/corr_stream.hip).
 * Why: the look-up gathers a 12 x 12 neighbourhood per (pixel, plane); row-major that is 12 partial lines (~2.1 KB moved for 400 B
 * used), tiled ~9 full lines.                                                                                                          */
int bflow_split_pack(const float* src, void* hi, void* lo, int R, int D, int N, int Np, bflow_stream_t stream);
int bflow_corr_build_split_tiled(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, float* out,
                                 int T, int B, int D, int h, int w, int Np, long long f1_target_stride, bflow_stream_t stream);
int bflow_corr_build_split(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, float* out,
                           int T, int B, int D, int N, int Np, long long f1_target_stride, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K4 / K9-K11  convolutions on the split-fp16 MFMA engine (csrc/conv_split.hip).
 * Replaces F.conv2d + norm + ReLU + residual of BasicEncoder / ResidualBlock (models/raft_utils/extractor.py:47-55,
 * 103-125) with an implicit GEMM over channels-last split tensors: two fp16 planes (hi, lo) of shape (B, H, W, C),
 * value = hi + lo * 2^-11 (fp32-class accuracy, ~2^-22 relative per product).
 *
 * bflow_conv_pack_weights : w (Cout, Cin, KH, KW) fp32 -> w_hi/w_lo (KH*KW*cin_pad/32, cout_pad, 32) fp16, zero padded;
 *                           cout_pad = multiple of the channel tile, cin_pad = multiple of 32.
 * bflow_conv_split        : out[b, ho, wo, co] = act( scale[co] * conv(x, w)[b, ho, wo, co] + shift[co] ), zero padding,
 *                           stride 1 or 2.  Outputs (any subset): fp32 NHWC `out_f32`, split NHWC `out_hi/out_lo`, both
 *                           addressed as (B, Ho*Wo, out_channel_stride) + out_channel_offset (concatenation for free);
 *                           `stats` (B, Cout, 2) fp64 receives += (sum, sum of squares) over Ho*Wo of the written
 *                           values (the InstanceNorm statistics; zero it first).                               */
typedef struct bflow_conv_desc {
    const void *x_hi, *x_lo;          /* blocked (B, C/32, P_in, 32) fp16 planes, C % 32 == 0                    */
    const void *w_hi, *w_lo;          /* packed weights (KH*KW*C/32, cout_pad, 32)                               */
    int B, H, W, C, Cout, cout_pad;
    int KH, KW, stride, pad_h, pad_w;
    int tile_n;                       /* output-channel tile: 64, 96 or 128 (cout_pad % tile_n == 0)             */
    float* out_f32;                   /* or NULL                                                                 */
    void *out_hi, *out_lo;            /* or NULL                                                                 */
    int out_channel_stride;           /* 0 = Cout                                                                */
    int out_channel_offset;
    int out_rows_per_image;           /* pixel rows per image in the output buffers; 0 = Ho*Wo (larger values leave
                                         zero tail rows, e.g. the 128-row tile padding K5 wants)                 */
    int in_rows_per_image;            /* pixel rows per image of x; 0 = H*W                                      */
    const void *x2_hi, *x2_lo;        /* optional second input: channels [x_split_channels, C) come from x2 (its own
                                         (B, (C - x_split_channels)/32, P_in, 32) planes): torch.cat for free        */
    int x_split_channels;             /* multiple of 32; ignored when x2 is NULL                                  */
    const float* addend;              /* optional blocked fp32 (B, ceil(Cout/32), P_out, 32) added before `act`
                                         (loop-invariant partial convolutions, update.py:35-37)                    */
    const float *scale, *shift;       /* per output channel, or NULL (= 1 / 0)                                   */
    int act;                          /* 0 = identity, 1 = relu, 2 = tanh                                        */
    double* stats;                    /* or NULL                                                                 */
    int gate;                         /* fused SepConvGRU gates (update.py:38-47), evaluated in the epilogue on v = conv + addend:
                                         0 = none;
                                         1 = "zr": Cout = 2*Ch.  channels [0, Ch): out_f32[c] = sigmoid(v) (the update gate z);
                                             channels [Ch, 2Ch): out_hi/lo[c - Ch] = split(sigmoid(v) * h[c - Ch])  (r * h);
                                         2 = "blend": Cout = Ch.  out_hi/lo[c] = split((1 - z[c]) * h[c] + z[c] * tanh(v));
                                         3 = "residual" (round 4; not a GRU gate, the same operand path): Cout = Ch.
                                             out_hi/lo[c] = split(relu(act(v) + h[c])): `relu(x + y)` of ResidualBlock.forward
                                             (extractor.py:55) with y = act(scale * conv + shift) = the folded-BatchNorm convolution and
                                             x = h; `act` may be 1 (relu) here;
                                         h = gate_h_hi/lo, z = gate_z; every gate buffer is blocked (B, Ch/32, P_out, 32) like the
                                         outputs (out_channel_stride = Ch, offset 0); out may alias h (blend, in place).       */
    const void *gate_h_hi, *gate_h_lo;
    const float* gate_z;
    int stats_replicas;               /* R: `stats` is (R, B, Cout, 2) and workgroup w adds into replica w % R -- thousands of fp64
                                         atomics on the same address serialise (measured: +20 % on the InstanceNorm convolutions);
                                         bflow_norm_act_split sums the replicas.  0 = 1.                                       */
    float* acc_nchw;                  /* optional fp32 (B, Cout, Ho*Wo): acc += result (after scale/shift/addend/act); the UPDATED
                                         value is what out_f32 / out_hi/lo receive.  Fuses BezierCurves.delta_update_params
                                         (bezier.py:137-139) and the re-emission of the Bezier channel block into the head's last
                                         convolution.                                                                          */
    int weight_sets;                  /* 0 / 1: one filter for every image.  S > 1: w_hi / w_lo hold S packed filters back to back and image
                                         b is multiplied with filter b % S (generic kernel).  Used by the weight-gradient GEMMs of the
                                         training path: "image" = (filter tap, k-chunk), "filter" = the packed output gradient of that
                                         k-chunk (bflow_wgrad_pack).                                                               */
    const float* x_raw;               /* optional: the input given as the PRE-NORMALISATION fp32 output of the previous convolution, blocked
                                         (B, C/32, P_in, 32), together with its InstanceNorm statistics `x_stats` (R, B, C, 2) fp64 as that
                                         convolution's epilogue accumulated them (x_stats_replicas = R, x_eps).  The kernel applies
                                         relu((x - mean) * rstd) -- extractor.py:47-48: relu(norm1(conv1(x))) -- while it stages the
                                         halo, with the coefficients bflow_norm_act_split would use, so the result equals the two-launch
                                         form bit for bit and the normalisation pass (a read and a write of the whole activation)
                                         disappears.  x_hi / x_lo are then ignored (may be NULL).  Stride-1 3x3, fp32 (+ stats) output only. */
    const double* x_stats;
    int x_stats_replicas;
    float x_eps;
    int keep_pad_channels;            /* split output with Cout % 32 != 0: the channels [Cout, next multiple of 32) of the last block are left
                                         untouched instead of written as zeros (they belong to another producer: the Bezier parameters that
                                         follow the motion features inside one block, update.py:95-97).  Cout % 4 == 0.                       */
} bflow_conv_desc_t;
/* bflow_conv_stem: the 7x7 stride-2 entry convolution of BasicEncoder (extractor.py:63,110) on a few-channel fp32 NCHW input
 * (5 / 8 / 25 / 41 / 3 channels): im2col in LDS over a TIGHT k = (channel, tap) index instead of 32-channel blocks per tap.
 *   x (B, Cin, H, W) fp32; weights: the (Cout, Cin, 7, 7) filter presented as a 1x1 convolution over K channels and packed with
 *   bflow_conv_pack_weights, where K is built chunk by chunk (chunks of min(Cin, 8) channels): per chunk the columns
 *   (c, r, q) -> c_local*49 + r*7 + q, zero padded to a multiple of 32; k_blocks = total K / 32.
 *   Outputs / epilogue as bflow_conv_split: blocked fp32 and/or split (B, ceil(Cout/32), out_rows_per_image, 32), per-channel
 *   scale / shift, act, InstanceNorm statistics.                                                                           */
typedef struct bflow_stem_desc {
    const float* x;
    const void *w_hi, *w_lo;
    int B, Cin, H, W, Cout, cout_pad, k_blocks;
    int ksize, stride, pad;           /* 7, 2, 3 */
    float* out_f32;
    void *out_hi, *out_lo;
    int out_rows_per_image;           /* 0 = Ho*Wo */
    const float *scale, *shift;
    int act;
    double* stats;
    int stats_replicas;               /* as in bflow_conv_desc_t */
    int n_windows;                    /* > 0: x is a wider source (B / n_windows, src_channels, H, W) and image n of the batch reads
                                         channels [window_starts[n / (B / n_windows)], + Cin) of source image n % (B / n_windows):
                                         gen_voxel_grids + torch.cat (raft.py:88-99,121) without materialising the stacked batch   */
    int src_channels;
    const int* window_starts;         /* host array, n_windows <= 8 entries */
    int layout;                       /* order of the packed filter's K axis (chunks of min(Cin, 8) channels):
                                         0: (chunk, c_local, r, q), each chunk zero padded to a multiple of 32 -- im2col tiles built in LDS;
                                         1: (chunk, r, window) with window = q * chunk + c_local zero padded to a multiple of 16 per filter row,
                                            each chunk's 7 rows zero padded to a multiple of 32 -- the row-window kernel (round 3): the input
                                            patch is split once and laid out [row][column][channel], a filter row's 7 x chunk values of an
                                            output pixel are contiguous and nothing is gathered.  The library picks the per-patch or the
                                            PERSISTENT form of that kernel (round 6: weights resident in LDS, wave-private windows, no
                                            barrier in the k-loop; fp32 + statistics output of a plain 5-channel input on >= 8192 slabs of
                                            2 x 16 pixels) -- same packed filter, bit-identical outputs;
                                         2 / 3: layout 1 with the persistent / the per-patch form FORCED (tests, A/B; 2 fails where the
                                            persistent form cannot run)                                                                      */
    /* General input of the row-window kernel (layout 1; all zero / NULL = the plain fp32 input above).  Replaces the torch element-wise and
     * concatenation launches of raft.py:131-140: `images = [2 * (x.float() / 255) - 1 ...]`, `fnet_img(images)` (the encoder's torch.cat of the
     * list, extractor.py:106-110) and `context_input = cat((context_grid, images[0]))`.
     *   window_bases   : host array of n_windows DEVICE pointers: window group g reads its channels from ITS OWN source tensor (the two images);
     *                    NULL: every window reads `x`;
     *   x2, x2_channels: the LAST x2_channels channels of every image come from image n % (B / n_windows) of x2 (x2_channels channels per
     *                    image); the window then supplies the first Cin - x2_channels;
     *   x_dtype / x2_dtype        : 0 fp32, 1 uint8 (element type of the window sources / of x2);
     *   x_image_norm / x2_image_norm: 1 = the element is 2 * (v / 255) - 1 (raft.py:134, the reference's operation order); zero padding is
     *                    applied to the normalised image.                                                                                   */
    const void* const* window_bases;
    const void* x2;
    int x2_channels;
    int x_dtype, x2_dtype, x_image_norm, x2_image_norm;
} bflow_stem_desc_t;
int bflow_conv_stem(const bflow_stem_desc_t* desc, bflow_stream_t stream);
int bflow_conv_pack_weights(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, int KH, int KW,
                            int cout_pad, int cin_pad, bflow_stream_t stream);
/* The filter of the input-gradient convolution packed straight from the FORWARD weight w (Cin, Cout, KH, KW) -- flipped in space, transposed in
 * channels: what autograd's conv_transpose of torch.nn.Conv2d uses -- (Cout, Cin = the roles of THIS convolution, i.e. the forward's Cin, Cout). */
int bflow_conv_pack_weights_adjoint(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, int KH, int KW,
                            int cout_pad, int cin_pad, bflow_stream_t stream);
int bflow_conv_split(const bflow_conv_desc_t* desc, bflow_stream_t stream);
/* Two INDEPENDENT convolutions of the same batch (neither reads what the other writes; disjoint outputs) issued on one stream.  When both
 * resolve to the same small-grid kernel that has a pair variant -- the generic split-k kernel (1x1 filters / im2col GEMMs on <= 320
 * workgroups) or the 8-wave 3x3 halo kernel -- they are ONE launch whose grid is the two grids back to back (the batch-1 motion encoder,
 * update.py:88-97: convc1 | convf1 and convc2 | convf2, which the reference leaves to the stream scheduler); otherwise two consecutive
 * launches.  The results are those of two bflow_conv_split calls bit for bit.  *fused (may be null) receives 1 for one launch, else 0.
 * BFLOW_CONV_NO_PAIR (environment, tools A/B) forces two launches. */
int bflow_conv_split_pair(const bflow_conv_desc_t* desc0, const bflow_conv_desc_t* desc1, int* fused, bflow_stream_t stream);

/* Training convolutions (bflow_amd/conv_train.py): the adjoints of Conv2d that the reference gets from autograd over
 * torch.nn.Conv2d (extractor.py / update.py convolutions) run on bflow_conv_split; these are the helpers around it.
 * bflow_wgrad_pack: re-blocks an NCHW fp32 tensor so that the PIXEL index becomes the contraction index of the conv engine:
 *     value(tap, k, c) = scale * src[b, c, yo*stride + r - pad_h, xo*stride + q - pad_w]   (0 outside the image / past the last pixel)
 *     with k = (b*Ho + yo)*Wo + xo, tap = r*KW + q;  split planes, taps_in_rows == 0: dst (KH*KW, k_blocks, rows, 32) [rows >= C],
 *     taps_in_rows != 0: dst (k_blocks, rows, 32) with row n = tap*C + c [rows >= KH*KW*C, the rest zeros].
 *   The weight gradient  dW[co, c, r, q] = sum_k dY[k, co] * X_tap[k, c]  is then a batch of 1x1 "convolutions" of the engine over k-chunks:
 *   one operand is the activation tensor (its rows = the GEMM's M), the other the packed filter of bflow_conv_desc_t.weight_sets
 *   ((k-tile, rows, 32) IS the engine's packed-weight layout).  `scale`: device pointer to one float, or NULL = 1.
 * bflow_blocked_f32_to_nchw: out[b, c, pix] = scale * x[b, c/32, pix, c%32]   (blocked fp32 engine output -> NCHW fp32).
 * bflow_pow2_scale: out2 = { s, 1/s } with s = 2^floor(log2(target / max|x|)) (1 for an all-zero / non-finite tensor), on the device;
 *   work8: 8 bytes, zero before the first call (the kernel leaves them zero).  Gradients of 1e-4..1e-9 are pre-scaled by s into fp16's
 *   normal range before they enter the split format and the results scaled back: exact in binary floating point.
 * bflow_grad_stats: out2 = {s, 1/s, s x C} (2 + C floats: the scale again once per channel, = scale_a of the bflow_norm_act_split pass that
 *   stages the gradient) for an NCHW gradient (B, C, HW) AND dbias[c] = sum over (b, pixel) -- the bias gradient autograd
 *   derives for torch.nn.Conv2d -- in one pass over x; partial: scratch of B*C*ceil(HW/1024) + 1024 floats (per-segment sums and
 *   per-workgroup maxima, combined in a fixed order by a second, one-workgroup launch: deterministic, no atomics).                 */
int bflow_wgrad_pack(const float* src, void* dst_hi, void* dst_lo, int B, int C, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                     int pad_h, int pad_w, int rows, int k_blocks, int taps_in_rows, const float* scale, bflow_stream_t stream);
int bflow_blocked_f32_to_nchw(const float* x, float* out, int B, int HW, int C, int channel_blocks, int rows_per_image, const float* scale,
                              bflow_stream_t stream);
int bflow_pow2_scale(const float* x, long long n, float target, float* out2, void* work8, bflow_stream_t stream);
/* bflow_rows_to_split: x (R, P, C) fp32 row-major [pixel][channel] -> blocked split (R, ceil(C/32), P, 32) = scale * x (pad channels zero): the
 *   engine's activation operand from a row-major matrix -- the adjoint of CorrComputation._corr_dot_prod_util (corr.py:264-272) contracts the
 *   volume gradient dC[n, m] over m for d f1 and over n for d f2; bflow_norm_act_split stages the second (NCHW-like) case, this the first.        */
int bflow_rows_to_split(const float* x, void* out_hi, void* out_lo, int R, int P, int C, const float* scale, bflow_stream_t stream);
int bflow_grad_stats(const float* x, int B, int C, int HW, float target, float* out2, float* partial, float* dbias, bflow_stream_t stream);
/* InstanceNorm2d / BatchNorm2d (+ ReLU) of the training path, forward and backward (extractor.py norm layers under autograd; csrc/norm_train.hip).
 * NCHW fp32; a "plane" is one (image, channel); mode 0 = instance (statistics per plane), 1 = batch (per channel over the batch, training mode).
 *   forward : bflow_plane_stats -> bflow_norm_train_finalize (mean, rstd, scale = gamma*rstd, shift = beta - mean*scale per plane; mode 1 also
 *             updates running_mean / running_var [unbiased] with `momentum` when they are given) -> bflow_norm_train_apply (y = [relu](x*scale+shift))
 *   backward: g = dy * [x*scale + shift > 0] (relu) | dy;  bflow_norm_train_bwd_stats (per plane sum g, sum g*xhat, fp64) ->
 *             bflow_norm_train_bwd_finalize (k1, k2 = their means over the normalisation set; mode 1: dgamma = sum g*xhat, dbeta = sum g) ->
 *             bflow_norm_train_bwd_apply (dx = scale * (g - k1 - xhat*k2)).                                                                 */
int bflow_norm_train_finalize(const double* stats, int mode, int B, int C, int HW, float eps, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float* mean, float* rstd, float* scale, float* shift,
                              bflow_stream_t stream);
int bflow_norm_train_apply(const float* x, const float* scale, const float* shift, float* y, long long planes, int HW, int relu, bflow_stream_t stream);
int bflow_norm_train_bwd_stats(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift,
                               double* sums, long long planes, int HW, int relu, bflow_stream_t stream);
int bflow_norm_train_bwd_finalize(const double* sums, int mode, int B, int C, int HW, float* k1, float* k2, float* dgamma, float* dbeta,
                                  bflow_stream_t stream);
int bflow_norm_train_bwd_apply(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift,
                               const float* k1, const float* k2, float* dx, long long planes, int HW, int relu, bflow_stream_t stream);
/* SepConvGRU gate arithmetic of the training path, forward and backward (update.py:33-48 under autograd; csrc/gru_gates.hip): fp32 NCHW,
 * C*HW % 4 == 0, 16-B aligned.  zr_pre (B, 2C, HW) = the z | r pre-activations of one merged convolution.
 *   bflow_gru_zr_fwd:    z = sigmoid(zr_pre[:, :C]), r = sigmoid(zr_pre[:, C:]), rh = r * h
 *   bflow_gru_zr_bwd:    dzr_pre = [dz * z(1-z) | drh * h * r(1-r)], dh = drh * r          (dz may be NULL = zeros)
 *   bflow_gru_blend_fwd: q = tanh(q_pre), h_new = (1 - z) * h + z * q
 *   bflow_gru_blend_bwd: dq_pre = dh_new * z * (1 - q^2), dz = dh_new * (q - h), dh = dh_new * (1 - z)                                  */
int bflow_gru_zr_fwd(const float* zr_pre, const float* h, float* z, float* r, float* rh, int B, int C, long long HW, bflow_stream_t stream);
int bflow_gru_zr_bwd(const float* dz, const float* drh, const float* z, const float* r, const float* h, float* dzr_pre, float* dh, int B, int C,
                     long long HW, bflow_stream_t stream);
int bflow_gru_blend_fwd(const float* q_pre, const float* z, const float* h, float* q, float* h_new, int B, int C, long long HW, bflow_stream_t stream);
int bflow_gru_blend_bwd(const float* dh_new, const float* q, const float* z, const float* h, float* dq_pre, float* dz, float* dh, int B, int C,
                        long long HW, bflow_stream_t stream);
/* bflow_wgrad_reduce: dw (Cout, Cin, KH*KW) = inv_scale * sum over the G k-chunks of the engine's blocked fp32 partial results:
 *   orientation 0: part (taps, G, blocks, rows >= Cin, 32), output channel = 32*block + lane;  orientation 1: part (G, blocks, rows >= Cout, 32),
 *   32*block + lane = tap*Cin + ci.                                                                                                  */
int bflow_wgrad_reduce(const float* part, float* dw, int G, int Cout, int Cin, int taps, int blocks, int rows, int orientation,
                       const float* inv_scale, bflow_stream_t stream);
/* bflow_conv_wgrad_halo: weight gradient of a stride-1 "same" convolution (3x3, 1x5, 5x1, 1x1) straight from the engine's blocked split
 * tensors -- X as the forward staged it, dY (pre-scaled) as the input-gradient pass staged it; nothing is re-packed:
 *     dw_acc[tap, co, ci] += sum_{b, y, x} dY[b, y, x, co] * X[b, y + r - KH/2, x + q - KW/2, ci]      (tap = r*KW + q)
 *   dw_acc: fp32 (KH*KW, ceil(Cout/64)*64, Cin_pad), zero before the call (fp32 atomics across the k-split workgroups);
 *   x_*: (B, Cin_pad/32, rows_per_image, 32), dy_*: (B, ceil(Cout/32), rows_per_image, 32).  The pixel contraction is fed by
 *   ds_read_b64_tr_b16 (transposed fragments from the row-major [pixel][channel] LDS tiles).
 * bflow_conv_wgrad_finish: dw (Cout, Cin, KH*KW) = inv_scale * dw_acc, and dw_acc is left ZERO (a persistent accumulator needs no memset). */
int bflow_conv_wgrad_finish(float* dw_acc, float* dw, int taps, int Cout, int Cin, int cin_pad, const float* inv_scale, bflow_stream_t stream);
int bflow_conv_wgrad_halo(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw_acc, int B, int H, int W, int Cin_pad,
                          int Cout, int rows_per_image, int KH, int KW, bflow_stream_t stream);

/* bflow_conv_thin_acc: the thin-output convolution of the Bezier head with its parameter update fused behind it:
 *     acc[b, co, y, x] += bias[co] + sum_{c, r, q} x[b, y+r-KH/2, x+q-KW/2, c] * w[co, c, r, q]      (zero padding, stride 1)
 * Replaces BezierHead.conv2 = Conv2d(256, 2*degree, 3, padding=1) (models/raft_spline/update.py:12-18) followed by
 * BezierCurves.delta_update_params (models/raft_spline/bezier.py:137-139).  Cout <= 32, 3x3 or 1x1 filters, C % 32 == 0, C <= 256.
 *   x_hi/x_lo : blocked split input (B, C/32, in_rows_per_image, 32);
 *   w_packed  : fp32 (KH*KW, Cout, C) -- plain fp32 weights, tap-major (no split: the kernel computes in fp32 on the vector ALU,
 *               the inputs are re-assembled as hi + lo * 2^-11);
 *   acc_nchw  : (B, Cout, H*W) fp32, updated in place;
 *   out_hi/lo : NULL, or a split buffer (B, out_channel_blocks, out_rows_per_image, 32) whose block `out_block` receives the
 *               UPDATED acc values at channels [out_channel_in_block, + Cout) -- the Bezier channels of the next GRU input.  With
 *               out_channel_in_block = 0 the other channels of the block are written as zeros; > 0: they are not touched (the
 *               parameters follow the motion features inside one block, exactly cat([out, bezier]) of update.py:95-97).      */
int bflow_conv_thin_acc(const void* x_hi, const void* x_lo, const float* w_packed, const float* bias, float* acc_nchw, void* out_hi,
                        void* out_lo, int B, int H, int W, int C, int in_rows_per_image, int Cout, int KH, int KW,
                        int out_channel_blocks, int out_block, int out_rows_per_image, int out_channel_in_block, bflow_stream_t stream);

/* bflow_conv_thin_mfma_acc: the SAME operation as bflow_conv_thin_acc (BezierHead.conv2 + delta_update_params, update.py:12-18,
 * bezier.py:137-139) for a 3 x 3 filter with Cout <= 28 (degree <= 14) on the matrix cores, "taps as output channels": a dense
 * 1 x 1 GEMM  Y[pixel][tap * Cout + co] = sum_c x[pixel][c] w[co][c][tap]  over a workgroup's 4 x 12 halo patch (fetched once by LDS-DMA;
 * 64 output rows per pass), then  out[p][co] = sum_tap Y[p + tap][tap * Cout + co]  through LDS.  Products are the engine's three-pass split products.
 *   w_hi/w_lo : the derived 1 x 1 filter W'[tap * Cout + co][c] = w[co][c][tap / 3][tap % 3] packed by bflow_conv_pack_weights
 *               (KH = KW = 1, Cout' = 9 * Cout, cout_pad >= Cout' rounded up to 64, cin_pad = C): (C/32, cout_pad, 32) fp16 planes;
 *   everything else as bflow_conv_thin_acc.                                                                                   */
int bflow_conv_thin_mfma_acc(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int cout_pad, const float* bias,
                             float* acc_nchw, void* out_hi, void* out_lo, int B, int H, int W, int C, int in_rows_per_image, int Cout,
                             int out_channel_blocks, int out_block, int out_rows_per_image, int out_channel_in_block, bflow_stream_t stream);

/* bflow_plane_stats: stats[p] = (sum, sum of squares) of plane p of an NCHW fp32 tensor (planes = B*C, HW % 4 == 0).
 * bflow_norm_act_split: out = act_out( res + act_a( norm_a(a) ) ) -> split NHWC (and/or fp32 NHWC), where
 *     a     : fp32 NHWC (B, HW, C), or NCHW (B, C, HW) when a_is_nchw (transposed on the fly);
 *     norm_a: InstanceNorm from stats_a (B, C, 2) [biased variance, eps inside the sqrt: F.instance_norm] if stats_a,
 *             else the per-channel affine scale_a/shift_a (folded BatchNorm), else identity;
 *     res   : nothing | split tensor res_hi/res_lo | InstanceNorm(b; stats_b) of a second fp32 NHWC tensor
 *             (the 1x1 down-sampling branch, extractor.py:43-44,52-53).                                         */
typedef struct bflow_norm_desc {
    const float* a; const double* stats_a; const float *scale_a, *shift_a; int a_is_nchw; int act_a;
    const float* b; const double* stats_b;
    const void *res_hi, *res_lo;
    int act_out;
    void *out_hi, *out_lo; float* out_f32;
    int B, HW, C; float eps;
    int rows_per_image;               /* pixel rows per image of the blocked tensors; 0 = HW                     */
    int stats_replicas;               /* stats_a / stats_b are (R, B, C, 2) tables to be summed; 0 = 1           */
    int act_b;                        /* 1: the b branch is relu(norm_b(b)) -- a residual relu(norm(conv)) that was never materialised
                                         (extractor.py:113 feeding the first block's `x + y`, extractor.py:55); 0: norm_b(b)               */
} bflow_norm_desc_t;
int bflow_plane_stats(const float* x, double* stats, long long planes, int HW, bflow_stream_t stream);
int bflow_norm_act_split(const bflow_norm_desc_t* desc, bflow_stream_t stream);

/* bflow_split_to_nchw: channels [c_first, c_first+c_count) of a split NHWC tensor -> fp32 (B, c_count, HW) planes with
 * an element batch stride (leaving the engine towards NCHW consumers).                                          */
int bflow_split_to_nchw(const void* x_hi, const void* x_lo, float* out, int B, int HW, int C, int c_first, int c_count,
                        long long out_batch_stride, bflow_stream_t stream);

/* Bezier parameter block on blocked tensors (P pixel rows per image).  The SepConvGRU gates and the per-iteration parameter update
 * are fused into bflow_conv_split's epilogue (`gate`, `acc_nchw`); what is left is the emission of the INITIAL parameter block:
 *   bflow_bezier_update    : params[b, c, pix] += delta[b, c, pix] (delta blocked fp32, first 2*deg channels) and the
 *                            updated parameters are re-emitted as one split channel block at block `cb_off` of a
 *                            (B, CB_total, P, 32) split buffer (the GRU input of the next iteration) and, if blk2 is
 *                            not NULL, of a second buffer.  delta may be NULL (only re-emit).  params: plain (B, C2, P) fp32 (the layout the look-up kernel reads).
 *                            channel_in_block > 0: the C2 values go to channels [channel_in_block, + C2) of the block and nothing else of it is written. */
int bflow_bezier_update(float* params, const float* delta, int C2, void* blk_hi, void* blk_lo, int CB_total, int cb_off,
                        void* blk2_hi, void* blk2_lo, int CB_total2, int cb_off2, int B, int P, int channel_in_block, bflow_stream_t stream);

/* bflow_im2col_small: out[b, pix, tap*C + c] = x[b, c, y+r-pad_h, x+q-pad_w] (zeros outside) as a blocked split tensor with
 *   ceil(KH*KW*C/32) channel blocks: turns the 7x7 convolution over the 2*deg Bezier channels (update.py:62,91) into a 1x1 GEMM. */
int bflow_im2col_small(const float* x, void* out_hi, void* out_lo, int B, int C, int H, int W, int KH, int KW, int pad_h, int pad_w,
                       int rows_per_image, bflow_stream_t stream);

/* K6  one pyramid level: 2x2 average pooling, stride 2, floor on odd sizes, over the target plane.
 * Replaces CorrData.get_downsampled (F.avg_pool2d), models/raft_utils/corr.py:108-125.
 *   in : (planes, h, w)   out : (planes, h/2, w/2)                                                    */
int bflow_corr_pool2x2(const float* in, float* out, long long planes, int h, int w, bflow_stream_t stream);

/* K7  9x9 bilinear window look-up in the correlation pyramid (zero padding, align_corners=True).
 * Replaces CorrBlockParallelMultiTarget.__call__, models/raft_utils/corr.py:307-351, and bilinear_sampler,
 * models/raft_utils/utils.py:5-21.  Plane p is one (pyramid level, target) pair, in the reference's channel
 * order (level-major, targets ascending inside a level); output channel = p*81 + (dy+4)*9 + (dx+4).   */
typedef struct bflow_plane {
    const float* base;   /* device: (B*N, h, w) slab of this level/target                               */
    int h, w;            /* plane size at this level                                                     */
    int level;           /* centroid = coords / 2^level                           (corr.py:333)          */
    int target;          /* index into the T base targets (which coords / which Bezier time to use)     */
} bflow_plane_t;

/*   planes : HOST array of P descriptors (copied into the kernel arguments)
 *   coords : (T, B, 2, h1, w1) pixel coordinates, channel 0 = x, 1 = y
 *   out    : (B, P*81, h1, w1)                                                                         */
int bflow_corr_lookup(const bflow_plane_t* planes, int P, const float* coords, float* out,
                      int T, int B, int h1, int w1, bflow_stream_t stream);

/* K8+K14+K7 fused: evaluates the Bezier curve at the look-up times inside the gather kernel, so neither
 * `flows` nor `coords1` (raft.py:180-181) is materialised:
 *   coords[t,b,:,y,x] = (x, y) + sum_i coef[t][i] * params[b, dim*deg + i, y, x]
 *   params : (B, 2*deg, h1, w1)            coef : HOST (T, deg) fp32 row-major (bflow_bezier_coeffs)    */
int bflow_corr_lookup_bezier(const bflow_plane_t* planes, int P, const float* params, const float* coef,
                             int T, int deg, float* out, int B, int h1, int w1, bflow_stream_t stream);
/* The same fused look-up writing the blocked split-fp16 tensor the conv engine consumes: planes (B, channel_blocks, rows_per_image, 32)
 * fp16, value = hi + lo * 2^-11, channel = plane * 81 + window index; channels >= P*81 of the last block are NOT written (keep the
 * buffer zero-initialised).  Removes the NCHW -> blocked conversion in front of BasicMotionEncoder.convc1 (update.py:88).            */
int bflow_corr_lookup_bezier_split(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                   void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                   bflow_stream_t stream);
/* ... on TILED planes (bflow_corr_build_split_tiled / bflow_corr_pool2x2_tiled): `base` of a descriptor points at (B*N, tiles*32) tiled
 * planes, h / w stay the LOGICAL plane size.  One workgroup = 8 query pixels x all planes, LDS-DMA gather of aligned 16-B units, zero
 * padding applied through the tap weights; every channel of the last channel block is written (pads as zeros).
 * bflow_corr_pool2x2_tiled: K6 (2x2 mean, floor on odd sizes) tiled -> tiled; pad positions of the output are zero.                    */
int bflow_corr_lookup_bezier_split_tiled(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                         void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                         bflow_stream_t stream);
int bflow_corr_pool2x2_tiled(const float* in, float* out, long long planes, int h, int w, bflow_stream_t stream);

/* fp16 correlation (BASELINE configs[4]: "fp16 MFMA correlation ... HBM-bound 4D volume stress").  The same three kernels on an fp16
 * volume: the build takes the PLAIN fp16 features (the hi planes of the split operands), runs ONE fp16 MFMA pass with fp32 accumulation
 * and stores fp16 tiled planes (half the bytes, a third of the matrix-core work; accuracy 2^-11 per operand and stored value instead of
 * 2^-22); pooling and look-up read / write fp16 planes (`base` of the descriptors points at fp16 data), the look-up output is unchanged.
 *   bflow_corr_build_f16_tiled : f1_hi (B | T*B, D/32, Np, 32), f2_hi (T*B, D/32, Np, 32) fp16 -> out (T, B, N, tiles*32) fp16; D in {128, 256} */
int bflow_corr_build_f16_tiled(const void* f1_hi, const void* f2_hi, void* out, int T, int B, int D, int h, int w, int Np,
                               long long f1_target_stride, bflow_stream_t stream);
int bflow_corr_pool2x2_tiled_f16(const void* in, void* out, long long planes, int h, int w, bflow_stream_t stream);

/* K5 with every arithmetic / storage combination of the streaming kernel (csrc/corr_stream.hip); replaces the same reference lines as
 * bflow_corr_build_split (models/raft_utils/corr.py:237-272).  Operands as bflow_corr_build_split_tiled; `out` (T, B, N, tiles*32).
 *   arithmetic 0: split pairs, three fp16 MFMA passes; f*_second = lo planes              (= bflow_corr_build_split_tiled when out_fp16 = 0)
 *              1: plain fp16 operands, one pass; f*_second ignored (may be null)           (= bflow_corr_build_f16_tiled when out_fp16 = 1)
 *              2: hi*hi on the fp16 rate + both cross terms of a 32-channel block in ONE v_mfma_f32_32x32x64_f8f6f4 (e4m3, unit scales);
 *                 f*_second = "x8" planes (rows, D/32, Np, 64 bytes) written by bflow_split_to_x8.  Two matrix-pipe units per product
 *                 instead of three; the cross terms carry 2^-11 of a product, their 2^-4 operand rounding leaves ~2^-16 per product.
 *   out_fp16    : the volume is stored as fp16 tiled planes (half the bytes) instead of fp32.
 * D in {128, 256}; 64 for (arithmetic 0, fp32 volume) only.
 * bflow_split_to_x8: hi, lo (rows, 32) fp16 planes of a split tensor -> x8 (rows, 64): [e4m3(hi) x 32 | e4m3(lo) x 32] per row (OCP e4m3,
 * round to nearest even, saturating at +-448; |x| < 2^-10 becomes 0: such elements keep fp16 accuracy in the product).
 *   pool_out    : optional (T1, B, N, tiles1*32) buffer of the volume's element type + pool_index (HOST array of T ints: row of target t in
 *                 pool_out, or -1): K6's level 0 -> 1 step (CorrData.get_downsampled, corr.py:108-125, as the pyramid constructor applies it,
 *                 corr.py:297-305) fused into the build: the 2 x 2 mean (F.avg_pool2d's summation order, floor on odd sizes, pad positions zero) of
 *                 the level-0 planes of the targets with more than one level is written by the same launch, every element of the level-1
 *                 planes included.  (arithmetic 0 | 2 with an fp32 volume) or (arithmetic 1 with an fp16 volume); D in {128, 256}; T <= 8.  */
int bflow_split_to_x8(const void* hi, const void* lo, void* x8, long long rows, bflow_stream_t stream);
int bflow_corr_build_tiled(const void* f1_hi, const void* f1_second, const void* f2_hi, const void* f2_second, void* out, int T, int B, int D,
                           int h, int w, int Np, long long f1_target_stride, int arithmetic, int out_fp16, void* pool_out, const int* pool_index,
                           bflow_stream_t stream);
int bflow_corr_lookup_bezier_split_tiled_f16(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                             void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                             bflow_stream_t stream);
/* bflow_corr_lookup_bezier_split_tiled (f16_planes = 0) / _tiled_f16 (1)  +  bflow_im2col_small(params, C = 2 deg, H = h1, W = w1, KH, KW, pads)
 * as ONE launch: the first workgroups of the look-up grid expand the filter windows of the same Bezier parameters into (col_hi, col_lo)
 * (B, ceil(KH KW 2 deg / 32), col_rows_per_image, 32) -- the two kernels an update iteration starts with (the look-up of raft.py:181-183 and the
 * input of the 7x7 `convf1`, update.py:91) read `params` and nothing of each other.  Outputs bit-identical to the two separate calls. */
int bflow_corr_lookup_im2col(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg, void* out_hi,
                             void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1, int f16_planes, void* col_hi,
                             void* col_lo, int KH, int KW, int pad_h, int pad_w, int col_rows_per_image, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K8  Bezier polynomial coefficients C(deg,i) (1-t)^(deg-i) t^i, i = 1..deg, computed in fp64 on the HOST
 * and rounded to fp32 exactly like BezierCurves._compute_flow_from_timestamps,
 * models/raft_spline/bezier.py:141-180.  times: HOST (T) fp64 in [0,1]; coef_out: HOST (T, deg) fp32.  */
int bflow_bezier_coeffs(const double* times, int T, int deg, float* coef_out);

/* K8  flow[t,b,d,y,x] = sum_i coef[t][i] * params[b, d*deg+i, y, x]  (+ (x,y) when add_coords0 != 0).
 * Replaces BezierCurves.get_flow_from_reference / einsum, bezier.py:185,188-216 and raft.py:181.
 *   out : (T, B, 2, h, w)                                                                              */
int bflow_bezier_eval(const float* params, const float* coef, int T, int deg, int B, int h, int w,
                      int add_coords0, float* out, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K13  convex up-sampling x8: softmax over the 9 taps of mask_scale*mask, 3x3 neighbourhood of 8*data,
 * pixel shuffle.  Replaces cvx_upsample, models/raft_utils/utils.py:33-48 (+ the 0.25 of update.py:125).
 * The mask logits are mask_scale * (mask + mask_bias[channel]); mask_bias (576) may be NULL.
 *   data : (B, C, h, w)   mask : (B, 576, h, w)   out : (B, C, 8h, 8w)                                  */
int bflow_cvx_upsample(const float* data, const float* mask, const float* mask_bias, float mask_scale,
                       float* out, int B, int C, int h, int w, bflow_stream_t stream);
/* The same operator reading the mask as the conv engine writes it: blocked fp32 (B, 18, mask_rows_per_image, 32) = the out_f32 of the mask
 * head's last convolution (update.py:120-125), bias included (no mask_bias here); equal masks give the bits of bflow_cvx_upsample. */
int bflow_cvx_upsample_blocked(const float* data, const float* mask_blocked, float mask_scale, float* out, int B, int C, int h, int w,
                               int mask_rows_per_image, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K1  event voxel grid: signed tri-linear (float x/y) or temporal-linear (integer x/y) accumulation of events.
 * Replaces VoxelGrid.convert, data/utils/representations.py:64-111.  `grid` (C,H,W) is WRITTEN WHOLE (no zero fill by the caller).
 * Tile-binned: the events are counted and placed per grid tile (two passes, deterministic prefix sums, no global atomics), every tile
 * is accumulated in LDS in 64-bit fixed point (2^-40 units) and stored once: the result is bit-identical from run to run and equals the
 * correctly rounded exact sum of the reference's fp32 contributions (|cell sum| < 2^23).  `workspace`: device memory, 16-byte
 * aligned, >= bflow_voxel_workspace_bytes(n_events, C, H, W, float_xy) bytes (returns -1 for an unsupported size), any contents.
 *   x,y : fp32 (f32xy), int16 (i16xy) or int32 (i32xy);  pol : int8 in {0,1};  t : int64 microseconds;  n_events <= 2^27.
 * Integer x/y follow the reference's FLAT index ht*wd*t + wd*y + x into Tensor.put_ (representations.py:85-94: only the time bin
 * is masked): an index in [-C*H*W, C*H*W) lands where put_ puts it (negative = from the end), one put_ would raise on is dropped;
 * no coordinate can cause an out-of-bounds write.                                                      */
long long bflow_voxel_workspace_bytes(long long n_events, int C, int H, int W, int float_xy);
int bflow_voxel_grid_f32xy(const float* x, const float* y, const signed char* pol, const long long* t,
                           long long n_events, long long t0_center, long long t1_center,
                           float* grid, int C, int H, int W, void* workspace, long long workspace_bytes, bflow_stream_t stream);
int bflow_voxel_grid_i16xy(const short* x, const short* y, const signed char* pol, const long long* t,
                           long long n_events, long long t0_center, long long t1_center,
                           float* grid, int C, int H, int W, void* workspace, long long workspace_bytes, bflow_stream_t stream);
int bflow_voxel_grid_i32xy(const int* x, const int* y, const signed char* pol, const long long* t,
                           long long n_events, long long t0_center, long long t1_center,
                           float* grid, int C, int H, int W, void* workspace, long long workspace_bytes, bflow_stream_t stream);

/* K2  in-place normalisation over the NON-ZERO entries: (v - mean) / std (unbiased), or v - mean if std == 0.
 * Replaces norm_voxel_grid, representations.py:9-18.  workspace: device, >= BFLOW_VOXEL_NORM_WS_DOUBLES doubles, any contents (round 6:
 * two launches -- per-block partial sums by plain stores, then fold + apply -- instead of a memset and three kernels).
 * bflow_voxel_merge_norm: the same over the concatenation [a | b] written to `out` (which may be `a`): the merge of the two per-window
 * grids of TwoStepSubSequence.__getitem__ (twostep.py:77-85: previous grid | current grid without its first bin) and its normalisation
 * in one read of each grid.                                                                                                            */
#define BFLOW_VOXEL_NORM_WS_DOUBLES 3072
int bflow_voxel_norm(float* grid, long long n, double* workspace, bflow_stream_t stream);
int bflow_voxel_merge_norm(const float* a, long long na, const float* b, long long nb, float* out, double* workspace, bflow_stream_t stream);

/* DSEC sample assembly: BaseSubSequence._rectify_events + _events_to_voxel_grid (data/dsec/subsequence/base.py:121-143)
 * inside K1's binning passes.  x, y: raw sensor coordinates (uint16, as stored in events.h5), pol: 0/1, t: int64 us;
 * rectify_map (H, W, 2) float32: rectify_map[y, x] = (x', y'), the rectified sub-pixel position, accumulated tri-linearly
 * like bflow_voxel_grid_f32xy (workspace: float_xy = 1).  Events with x >= W or y >= H are skipped and counted in *bad_count
 * (may be NULL; the reference asserts on them).
 * bflow_maxabs_diff: *out = max(*out, max_i |a[i] - b[i]|) (out zeroed by the caller) -- the agreement check of the temporal
 * slice shared by the previous and the current grid (data/dsec/subsequence/twostep.py:83).                                  */
int bflow_voxel_grid_rectified(const unsigned short* x, const unsigned short* y, const unsigned char* pol, const long long* t,
                               long long n, const float* rectify_map, long long t0_center, long long t1_center, float* grid,
                               int C, int H, int W, int* bad_count, void* workspace, long long workspace_bytes, bflow_stream_t stream);
/* The same four launches with the WINDOW taken from device memory at run time, so that one captured hipGraph serves every frame of a
 * recording (bflow_amd/pipeline.py EventFrameGraph: the assembly of frame k + 1 as a branch of the graph that runs frame k's forward).
 *   x, y, pol, t : the whole time-sorted recording (n_total events, resident);
 *   window       : device, 4 int64, 8-byte aligned: {first event, event count, t0_center, t1_center} -- what
 *                  BaseSubSequence._get_events / construct_voxel_grid derive on the host (data/dsec/subsequence/base.py:160-204:
 *                  np.searchsorted of the extended time window, centres = ts_from / ts_to), written by the caller before each replay;
 *   max_events   : the capacity the launches and the workspace (bflow_voxel_workspace_bytes(max_events, C, H, W, 1)) are planned for.
 * The kernels clamp the window to [0, n_total) and to max_events (the host side asserts count <= max_events: a clamped window would
 * silently drop events).  Results are bit-identical to bflow_voxel_grid_rectified on the same window (fixed-point accumulation does not
 * depend on the chunking).                                                                                                          */
int bflow_voxel_grid_rectified_window(const unsigned short* x, const unsigned short* y, const unsigned char* pol, const long long* t,
                                      long long n_total, long long max_events, const long long* window, const float* rectify_map,
                                      float* grid, int C, int H, int W, int* bad_count, void* workspace, long long workspace_bytes,
                                      bflow_stream_t stream);
int bflow_maxabs_diff(const float* a, const float* b, long long n, float* out, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K15 end-point-error partial sums for epe_masked / the EPE metric state, utils/metrics.py:30-49,196-213:
 *   acc[0] += sum over valid pixels of sqrt(sum_c (pred-gt)^2)   (fp64)
 *   acc[1] += number of valid pixels                              (fp64)
 *   pred, gt : (B, C, HW);  valid : (B, HW) uint8 or NULL (= all valid);  acc : device double[2], the caller
 *   zeroes it.  The caller forms mean = acc[0]/acc[1] (per batch, then sums batch means like the metric). */
int bflow_epe_accumulate(const float* pred, const float* gt, const unsigned char* valid,
                         int B, int C, long long HW, double* acc, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Validation harness: the per-batch sums behind AE / NPE / EPE_MULTI / AE_MULTI (utils/metrics.py:51-193,
 * 216-296), EPE_MULTI's trajectory-length filter (:60-64) and InputPadder.pad (modules/utils.py:48-83).
 *   bflow_flow_metrics_accumulate: one pass over (pred, gt, valid) as in bflow_epe_accumulate; acc is device double[6], zeroed
 *       by the caller: acc[0] += sum epe, acc[1] += valid pixels, acc[2] += sum of the angular error in RADIANS between
 *       (u, v, 1) vectors (cosine clamped to [-1, 1]), acc[3+k] += #pixels with epe > n_pixels_k and epe / max(|gt|, 1e-6) >= 0.05.
 *   bflow_traj_len: targets (M, B, C, HW) stacked ground-truth flows -> out (B, HW) = sum_m ||t_m - t_{m-1}||_2.
 *   bflow_pad_replicate: x (planes, H, W) -> out (planes, H + top + bottom, W + left + right), edge values replicated.       */
int bflow_flow_metrics_accumulate(const float* pred, const float* gt, const unsigned char* valid, int B, int C, long long HW,
                                  float n_pixels0, float n_pixels1, float n_pixels2, double* acc, bflow_stream_t stream);
int bflow_traj_len(const float* targets, float* out, int M, int B, int C, long long HW, bflow_stream_t stream);
int bflow_pad_replicate(const float* x, float* out, long long planes, int H, int W, int pad_left, int pad_right, int pad_top,
                        int pad_bottom, bflow_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Training path: adjoints of K7 / K6 / K13 and the masked L1 of the sequence losses.  The reference gets them
 * from autograd over F.grid_sample (raft_utils/utils.py:5-21, corr.py:307-351), F.avg_pool2d (corr.py:108-125), unfold + softmax
 * (utils.py:33-48) and abs/sum/mean (utils/losses.py:6-22).  All deterministic (no fp atomics on shared addresses).
 *   bflow_corr_lookup(_bezier)_bwd: planes = the FORWARD plane table; grad_planes = HOST array of P device pointers, the gradient of
 *       every plane (same shape as the plane), ACCUMULATED in place (the caller zeroes it once per step: the volume gradient sums
 *       over all GRU iterations); grad_out (B, P*81, h1, w1); grad_coords (P, B, 2, h1, w1) is WRITTEN per plane -- the caller sums
 *       the planes of a target (and applies the Bezier coefficients for the fused variant).
 *   bflow_corr_pool2x2_bwd: grad_prev (M, h, w) += 0.25 * grad_cur (M, h/2, w/2) over the 2x2 cells (odd tail row/column: none).
 *   bflow_cvx_upsample_bwd: grad_up (B, C, 8h, 8w), data (B, C, h, w), mask (B, 576, h, w) logits as given to bflow_cvx_upsample with
 *       mask_bias = NULL, mask_scale = 1 -> grad_data (B, C, h, w), grad_mask (B, 576, h, w); scratch: B*C*72*h*w floats.
 *   bflow_l1_masked_accumulate: acc[0] += sum over valid positions of sum_c |src - tgt|, acc[1] += #valid positions (valid NULL = all);
 *       src, tgt (B, C, HW); acc device double[2] zeroed by the caller; loss = acc[0] / acc[1]  (losses.py:12-22).
 *   bflow_l1_masked_grad: grad (B, C, HW) = weight * upstream[0] / acc[1] * sign(src - tgt) on valid positions, 0 elsewhere
 *       (upstream: device float[1] or NULL = 1).                                                                                 */
int bflow_corr_lookup_bwd(const bflow_plane_t* planes, float* const* grad_planes, int P, const float* coords, int T,
                          const float* grad_out, float* grad_coords, int B, int h1, int w1, bflow_stream_t stream);
int bflow_corr_lookup_bezier_bwd(const bflow_plane_t* planes, float* const* grad_planes, int P, const float* params, const float* coef,
                                 int T, int deg, const float* grad_out, float* grad_coords, int B, int h1, int w1,
                                 bflow_stream_t stream);
int bflow_corr_pool2x2_bwd(const float* grad_cur, float* grad_prev, long long M, int h, int w, bflow_stream_t stream);
int bflow_cvx_upsample_bwd(const float* grad_up, const float* data, const float* mask, float* grad_data, float* grad_mask,
                           float* scratch, int B, int C, int h, int w, bflow_stream_t stream);
int bflow_l1_masked_accumulate(const float* src, const float* tgt, const unsigned char* valid, int B, int C, long long HW,
                               double* acc, bflow_stream_t stream);
int bflow_l1_masked_grad(const float* src, const float* tgt, const unsigned char* valid, int B, int C, long long HW,
                         const double* acc, const float* upstream, float weight, float* grad, bflow_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* BFLOW_HIP_H */
