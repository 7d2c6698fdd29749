/*
 * snap_hip.h -- C ABI of libsnap_hip.so: the MI355X (gfx950) kernels behind the
 * SNAP BEV-fusion + pose-matching hot path.
 *
 * The reference (google-research/snap) is pure Python/JAX and exposes NO FFI /
 * plugin ABI; its boundary is the Flax module API.  This
 * header is therefore the build-defined boundary a JAX custom-call / XLA FFI
 * handler (or ctypes, as snap_amd/_lib.py does) would bind.  Each entry point
 * cites the reference expression (file:line under /root/reference) it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; no torch / C++ types; the caller owns every buffer
 *     and pre-allocates outputs; the only hidden state is none (no allocation).
 *   - tensors are row-major, channels-last, contiguous; float = IEEE fp32;
 *     masks are uint8 (0/1); indices are int32.
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no internal sync;
 *     re-entrant and thread-safe for distinct streams.
 *   - returns SNAP_OK (0) or a negative SnapStatus; never throws across the ABI.
 */
#ifndef SNAP_HIP_H_
#define SNAP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum SnapStatus {
  SNAP_OK = 0,
  SNAP_ERR_BAD_SHAPE = -1,    /* inconsistent / unsupported sizes */
  SNAP_ERR_UNSUPPORTED = -2,  /* option not implemented */
  SNAP_ERR_NULL = -3,         /* required pointer is NULL */
  SNAP_ERR_LAUNCH = -4,       /* hipGetLastError() != hipSuccess after launch */
  SNAP_ERR_WORKSPACE = -5     /* workspace too small */
} SnapStatus;

/* Library / device introspection. */
int snap_abi_version(void);                 /* bumps on any signature change */
const char* snap_status_string(int status);
const char* snap_build_arch(void);          /* "gfx950" */

/* ------------------------------------------------------------------------- *
 * Encoder: implicit-GEMM convolution engine on f32 MFMA (v_mfma_f32_32x32x2_f32)
 *   replaces flax.linen.Conv / StdConv / Dense as used by
 *   snap/models/resnet.py:73-132,183-216, snap/models/image_encoder.py:53-94,
 *   snap/models/layers.py:55-78 (MLP Dense stack) and, with a (H x W x D)-sized
 *   kernel, the direct rotated-template correlation of
 *   snap/models/pose_exhaustive_voting.py:86-91.
 * ------------------------------------------------------------------------- */
enum { SNAP_PRO_NONE = 0, SNAP_PRO_AFFINE = 1, SNAP_PRO_GN_RELU = 2,
       SNAP_PRO_RELU_GN = 3, SNAP_PRO_RELU = 4 };
enum { SNAP_EPI_BIAS = 1, SNAP_EPI_RELU = 2, SNAP_EPI_RESIDUAL = 4,
       SNAP_EPI_UPSAMPLE2X_ADD = 8, SNAP_EPI_ROWMASK = 16,
       SNAP_EPI_GELU = 32 /* tanh-approximated GELU (ViT MLP), applied where ReLU is */ };

typedef struct SnapConvDesc {
  int32_t N, H, W, Cin, Cin_stride;   /* input  x[N,H,W,Cin_stride], first Cin used */
  int32_t KH, KW, stride, pad_t, pad_l;
  int32_t Ho, Wo, Cout, Cout_stride;  /* output y[N,Ho,Wo,Cout_stride]            */
  int32_t prologue;                   /* SNAP_PRO_*  applied to every input element
                                         BEFORE zero padding                       */
  int32_t epilogue;                   /* OR of SNAP_EPI_*                          */
  float in_scale, in_shift;           /* SNAP_PRO_AFFINE: x*in_scale + in_shift    */
  int32_t tile_hint;                  /* 0 = the engine picks the output tile; bm * 1000 + bn
                                         (128128 | 128064 | 64128 | 64064) forces one -- part of
                                         the descriptor because the size queries depend on it
                                         (tests and tuning tools; the library reads no
                                         environment variables).  + 1 000 000 x mode selects
                                         among the 1x1 bodies of the split engine (conv_rs.hip):
                                         0 automatic, 1 tiled body only, 2 the stationary
                                         kernels also below their row-count threshold, 3 no
                                         weights-stationary kernel, 4 = 2 and 3              */
} SnapConvDesc;

/* y = epilogue( conv( prologue(x), w ) ).  w is HWIO flattened: [KH*KW*Cin, Cout].
 * gn_mu/gn_sc: [N, Cin] per-(image, channel) mean and rstd*gamma (from
 * snap_group_norm_stats_f32); gn_beta: [Cin].  residual: same layout as y.
 * up_prev: [N, Ho/2, Wo/2, Cout_stride] (bilinear x2, half-pixel centres, edge
 * clamp == jax.image.resize, image_encoder.py:90).  row_mask: [N*Ho*Wo] uint8.
 * Unused pointers may be NULL. */
int snap_conv2d_nhwc_f32(const SnapConvDesc* desc, const float* x, const float* w,
                         float* y, const float* gn_mu, const float* gn_sc,
                         const float* gn_beta, const float* bias,
                         const float* residual, const float* up_prev,
                         const uint8_t* row_mask, void* stream);

/* Optional extras of a conv launch (all may be NULL / 0):
 *  - row-indexed launch over a masked voxel list (the fusion MLP of
 *    streetview_encoder.py:281 runs over every voxel and the result is then masked by
 *    visibility; unobserved voxels never reach any output, so only the observed rows are
 *    multiplied).  rows_in: GEMM row m reads output pixel rows_in[m] of x; rows_out: GEMM
 *    row m is stored to row rows_out[m] of y; row_count: DEVICE scalar with the number of
 *    rows (<= N*Ho*Wo, the launch bound) -- no host synchronisation.  Only
 *    SNAP_EPI_BIAS / SNAP_EPI_RELU epilogues.
 *  - gn_partial: the epilogue also emits per-(image, row tile, channel) sums of y and y^2
 *    (of relu(y) if gn_partial_relu), from which snap_group_norm_stats_from_partial_f32
 *    produces the GroupNorm statistics the NEXT layer's fused prologue needs
 *    (resnet.py:46-60) without re-reading y.  Size: snap_conv2d_gn_partial_bytes(desc);
 *    0 means "not available for this shape" (Ho*Wo smaller than a row tile); a launch that
 *    emits statistics never splits K.  */
typedef struct SnapConvExtras {
  const int32_t* rows_in;
  const int32_t* rows_out;
  const int32_t* row_count;
  float* gn_partial;
  size_t gn_partial_bytes;
  int32_t gn_partial_relu;
  void* workspace;            /* split-K scratch: snap_conv2d_workspace_bytes(desc) (may be 0) */
  size_t workspace_bytes;
  const void* w_bf16;         /* non-NULL selects a bf16 matrix-core engine (see below) */
  size_t w_bf16_bytes;
  int32_t w_split_parts;      /* 0: w_bf16 = rounded weights (training-precision engine);
                                 2 | 3: w_bf16 = split weights (f32-grade split-bf16 engine) */
  int32_t w_split_root;       /* 1: the 7 x 7 / stride 2 / pad 3 root convolution (resnet.py:200-205)
                                 of an RGB image stored with 4 floats per pixel (Cin = 3,
                                 Cin_stride = 4): w_bf16 = snap_conv2d_pack_weights_split_root_bf16,
                                 a K slab = 4 consecutive pixels of one kernel row */
  float* gn_partial2;         /* with gn_partial (gn_partial_relu = 0): a second buffer of the same
                                 size that MAY receive the partial sums of relu(y) -- for an output
                                 read by a GroupNorm->ReLU layer AND a ReLU->GroupNorm layer (the
                                 last unit of a ResNet stage: next stage / FPN level).  A request,
                                 honoured by one kernel variant (split-bf16 engine, GroupNorm->ReLU
                                 prologue, 128 x 128 tiles): */
  size_t gn_partial2_bytes;
  int32_t gn_partial2_done;   /* OUT: 1 if gn_partial2 was written by this launch, else 0 (take
                                 snap_group_norm_stats_f32 for the second statistic) */
  int32_t x_presplit;         /* 1: `x` is NOT f32 but the pre-split image of the (already normalised)
                                 input, [N*H*W][Cin/16][hi 16 | lo 16] bf16 as written by
                                 snap_gn_norm_split_f32 / snap_presplit_f32; needs w_split_parts = 2,
                                 prologue NONE, Cin % 16 == 0, no row lists.  Both operands then
                                 travel by LDS-DMA (conv_ps.hip); sizes come from the
                                 snap_conv2d_presplit_* queries below instead of the plain ones.
                                 With w_split_parts = 1 (w_bf16 = the ONE-part image of
                                 snap_conv2d_pack_weights_split_bf16): `x` is a plain bf16 tensor
                                 [N,H,W,Cin] (Cin_stride == Cin) and the launch is the
                                 training-precision arithmetic (operands rounded to bf16, f32
                                 accumulate: the bits of the x_half engine) on the same ring; no
                                 statistics / split-K / up-sampling epilogue; y_half allowed */
  int32_t ps_tile;            /* ... 0 = automatic, 1 = 128-row tiles, 2 = 256-row tiles (tuning) */
  int32_t ps_res_init;        /* ... 1 = a residual is loaded into the accumulators before the K loop
                                 (r + p1 + p2 + ... instead of (p1 + p2 + ...) + r) */
  int32_t bk_hint;            /* f32 engine: K-slab depth 16 | 32 of the large tiles (0 = default 16) */
  int32_t tune_flags;         /* SNAP_TUNE_*: A/B switches for tests and tuning tools (0 = defaults) */
  int32_t gn_partial_rows;    /* 0: gn_partial is laid out per row tile of the launch
                                 (snap_conv2d_tile_rows_ex); 32: per 32-row slab, sized by
                                 snap_conv2d_splitk_gn_partial_bytes -- a split-K launch of the
                                 split-operand engine (workspace given), whose reduce pass then emits
                                 the partial sums (a split-K launch has no epilogue that sees
                                 finished outputs) */
  int32_t w_half;             /* with w_split_parts = 0: 1 = the training-precision engine in IEEE
                                 half -- w_bf16 holds snap_conv2d_pack_weights[_multi]_f16 images,
                                 the A operand is rounded to f16 (RNE; beyond 65504 -> inf) and the
                                 products run on v_mfma_f32_32x32x16_f16 (f32 accumulate): the
                                 reference's dtype=float16 train config (train_localization.py:93,
                                 resnet.py:97), driven by DynamicScale (trainer.py:391-392) */
  int32_t x_half;             /* with w_split_parts = 0 (training-precision engine), prologue NONE, no row
                                 lists, Cin_stride % 8 == 0: 1 = `x` is NOT f32 but the tensor already
                                 rounded to the engine's element type (bf16, or IEEE half with w_half),
                                 [N, H, W, Cin_stride] -- the half-precision twin a GroupNorm VJP wrote
                                 next to its f32 gradient (snap_group_norm_bwd_ex_f32).  The
                                 data-gradient convolution then moves both operands by LDS-DMA: half the
                                 input bytes, no conversion in the loop; same bits as the f32 input.
                                 rows_out / row_count are honoured (compact row buffers), rows_in is not */
  void* y_half;               /* with w_split_parts = 0, prologue NONE / RELU, no statistics: the stored
                                 values ALSO go, rounded (RNE) to the engine's element type, to y_half
                                 (same indexing as y, 2-byte elements) -- and ONLY there when y is NULL:
                                 the hidden activations / inter-layer gradients of the masked MLP
                                 (layers.py:55-78), which every consumer rounds to that type anyway */
  /* The statistics pass of a GroupNorm VJP inside the data-gradient convolution that produces its incoming
   * gradient (x_half launches without split-K, Cout_stride == Cout, no row lists; gn_partial / gn_partial_bytes
   * as for the forward statistics: snap_conv2d_gn_partial_bytes_ex(desc, 0)).  gnb_mode = SNAP_PRO_GN_RELU /
   * SNAP_PRO_RELU_GN (0 = off): with dz = the stored output, xh = (x - mu) * rstd (relu(x) for RELU_GN) and
   * dyp = dz gated by (xh * gamma + beta > 0) (GN_RELU) the epilogue emits, per (image, row tile, channel), the
   * sums of dyp and dyp * xh -- what snap_group_norm_bwd_ex_f32's first pass computes by re-reading x and dz;
   * snap_group_norm_bwd_stats_f32 consumes them.  gnb_x [N, Ho, Wo, Cout] f32 (the layer input the GroupNorm
   * normalises), gnb_mu / gnb_rstd [N, Cout], gnb_gamma / gnb_beta [Cout]. */
  const float* gnb_x;
  const float* gnb_mu;
  const float* gnb_rstd;
  const float* gnb_gamma;
  const float* gnb_beta;
  int32_t gnb_mode;
} SnapConvExtras;
#define SNAP_TUNE_NO_HALO 1   /* split engine: the im2col body for every 3x3 convolution */
#define SNAP_TUNE_RAW_RING 2   /* split engine: the raw-row LDS-DMA ring body (conv_raw.hip) for the K >= 256 1x1 layers with a GroupNorm prologue -- same bits as the tiled body; opt-in (measured level to 20 % slower: DESIGN.md 5a) */
#define SNAP_TUNE_NO_PLAIN 8   /* split engine: the general A loader also for 1x1 / stride-1 / unpadded layers */
#define SNAP_TUNE_RS_NSPLIT_SHIFT 4   /* bits 4..7: row-stationary kernel, forced column split (0 = automatic) */
/* Pre-split launches (extras->x_presplit): row tile, GroupNorm partial-sum bytes and split-K
 * workspace bytes of the launch `desc` + `ps_tile` describes (the counterparts of
 * snap_conv2d_tile_rows / _gn_partial_bytes / _workspace_bytes). */
int32_t snap_conv2d_presplit_tile_rows(const SnapConvDesc* desc, int32_t ps_tile);
size_t snap_conv2d_presplit_gn_partial_bytes(const SnapConvDesc* desc, int32_t ps_tile);
size_t snap_conv2d_presplit_workspace_bytes(const SnapConvDesc* desc, int32_t ps_tile);
/* 1 when the pre-split engine takes the shape `desc` describes (prologue NONE, Cin % 16 == 0, the
 * input window of a row tile and one column tile of the weight image below its 32-bit offset
 * limits), else 0: callers that have another engine for the shape (the exhaustive voting of
 * pose_exhaustive_voting.py:72-104 on very large templates) ask before they pre-split. */
int32_t snap_conv2d_presplit_supported(const SnapConvDesc* desc);
/* GroupNorm -> ReLU (resnet.py:34-60,117-130) of a conv output y [N, HW, C] whose partial sums
 * came out of the producing conv's epilogue (extras->gn_partial, row tile `tile_rows`), written
 * ONCE in the pre-split format: out [N*HW][C/16][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] bf16
 * (N*HW*C*4 bytes), hi = bf16(v), lo = bf16(v - hi).  Replaces the statistics finalize launch of
 * that tensor AND the normalise / split work of every consumer tile; mu / sc (optional, both or
 * neither): the statistics as snap_group_norm_stats_from_partial_f32 defines them.  C % 16 == 0,
 * C <= 2048, groups a power of two <= 64. */
int snap_gn_norm_split_f32(const float* y, const float* partial, int32_t N, int32_t HW, int32_t C,
                           int32_t groups, float eps, int32_t tile_rows, const float* gamma,
                           const float* beta, void* out, float* mu, float* sc, void* stream);
/* The plain two-part split of x [rows, C] f32 into the same format (C % 16 == 0). */
int snap_presplit_f32(const float* x, int64_t rows, int32_t C, void* out, void* stream);

int snap_conv2d_nhwc_ex_f32(const SnapConvDesc* desc, const float* x, const float* w,
                            float* y, const float* gn_mu, const float* gn_sc,
                            const float* gn_beta, const float* bias,
                            const float* residual, const float* up_prev,
                            const uint8_t* row_mask, const SnapConvExtras* extras,
                            void* stream);
/* Training-precision engine: the analogue of the reference's float16 train config
 * (snap/configs/train_localization.py:25).  Tensors stay f32 in memory; the prologue runs in
 * f32, then both operands are rounded to bf16 (round-to-nearest-even) and multiplied on the
 * bf16 matrix cores with f32 accumulation; epilogue in f32.  The caller packs the weights
 * once per launch (or per step) with snap_conv2d_pack_weights_bf16 -- out is
 * [Cout][taps][roundup(Cin, 8)] bf16, taps = KH*KW -- and passes them as extras->w_bf16; `w`
 * is still required (shapes the bf16 engine does not carry -- Cin < 4, unaligned channel
 * rows -- run on the exact f32 engine instead).  All fusions, row-indexed launches, GroupNorm
 * partial sums and split-K behave as on the f32 engine. */
size_t snap_conv2d_packed_weights_split_root_bytes(int32_t Cout, int32_t parts);
int snap_conv2d_pack_weights_split_root_bf16(const float* w /* [7,7,3,Cout] */, int32_t Cout,
                                             int32_t parts, void* out, size_t out_bytes,
                                             void* stream);
size_t snap_conv2d_packed_weights_bytes(int32_t taps, int32_t Cin, int32_t Cout);
int snap_conv2d_pack_weights_bf16(const float* w, int32_t taps, int32_t Cin, int32_t Cout,
                                  void* out, size_t out_bytes, void* stream);
/* ... the same image in IEEE half for extras->w_half = 1 (the reference's float16 compute /
 * parameter dtype, train_localization.py:93, resnet.py:97): same size, same layout */
int snap_conv2d_pack_weights_f16(const float* w, int32_t taps, int32_t Cin, int32_t Cout,
                                 void* out, size_t out_bytes, void* stream);
/* f32-grade engine on the bf16 matrix cores (inference / parity path; replaces the same
 * flax.linen.Conv / Dense calls: resnet.py:73-132, image_encoder.py:67-94, layers.py:55-78).
 * Each f32 operand is split into `parts` bf16 values (hi = bf16(v), then bf16 of the exact f32
 * residual, ...); a*b is the f32-accumulated sum of the part products above the f32 rounding
 * level: parts = 2 -> a_lo b_hi + a_hi b_lo + a_hi b_hi (relative error ~2^-17 per product),
 * parts = 3 -> six products (operands exact to 24 bits, ~2^-24 per product: the accuracy class
 * of an f32 fmaf chain).  The caller splits the weights with
 * snap_conv2d_pack_weights_split_bf16 -- out is the engine's own tile-major image
 * ([Cout/128][taps][Cin/16][parts][128][16] bf16, zero padded; an opaque blob to the caller) --
 * and passes them as extras->w_bf16 with extras->w_split_parts = parts; activations are split
 * on the fly after the f32 prologue.  Fusions, row-indexed launches, GroupNorm partial sums
 * and split-K as on the f32 engine; shapes it does not carry (Cin < 4, unaligned channel rows)
 * run on the exact f32 engine. */
size_t snap_conv2d_packed_weights_split_bytes(int32_t taps, int32_t Cin, int32_t Cout,
                                              int32_t parts);
int snap_conv2d_pack_weights_split_bf16(const float* w, int32_t taps, int32_t Cin, int32_t Cout,
                                        int32_t parts, void* out, size_t out_bytes, void* stream);
/* ... every kernel of an encoder in ONE launch: `items` is a DEVICE array sorted by block_begin;
 * item i owns workgroups [block_begin, block_begin + snap_conv2d_pack_weights_split_blocks(...));
 * out: snap_conv2d_packed_weights_split_bytes(taps, Cin, Cout, parts) bytes, 16-byte aligned. */
typedef struct SnapPackItem {
  const float* w;
  void* out;
  int32_t taps, Cin, Cout, block_begin;
} SnapPackItem;
int32_t snap_conv2d_pack_weights_split_blocks(int32_t taps, int32_t Cin, int32_t Cout);
/* The same for the bf16-operand engine's images (training precision): one launch for every kernel
 * of a step.  An item with taps < 0 asks for the ROTATED image of |taps| taps that the
 * data-gradient convolution reads -- the image of w.flip(0, 1).permute(0, 1, 3, 2) with the output
 * channels padded to a multiple of four: out = snap_conv2d_packed_weights_bytes(|taps|, Cout,
 * roundup(Cin, 4)) bytes.  Items sorted by block_begin; item i owns
 * snap_conv2d_pack_weights_blocks(taps, Cin, Cout) workgroups. */
int32_t snap_conv2d_pack_weights_blocks(int32_t taps, int32_t Cin, int32_t Cout);
int snap_conv2d_pack_weights_multi_bf16(const SnapPackItem* items, int32_t n_items,
                                        int32_t total_blocks, void* stream);
/* ... the same images in IEEE half (SnapConvExtras.w_half; same sizes and layout) */
int snap_conv2d_pack_weights_multi_f16(const SnapPackItem* items, int32_t n_items,
                                       int32_t total_blocks, void* stream);
int snap_conv2d_pack_weights_split_multi_bf16(const SnapPackItem* items, int32_t n_items,
                                              int32_t total_blocks, int32_t parts, void* stream);

/* Generic N-D grid operators (snap/utils/grids.py:116-153); n = 1..3 leading grid axes.
 * snap_interpolate_nd_f32: array [size..., D], points [K, n] in corner-origin coordinates (cell
 * centres at k + 0.5), optional valid_array [size...] -> values [K, D], valid [K]:
 * jax.scipy.ndimage.map_coordinates(order=1, mode='nearest') per channel after the -0.5 shift
 * (:129-130); valid = 0 <= p < size on every axis AND no tap -- not even a zero-weight one --
 * is invalid (the 0 * NaN mask of :131-136).  `size` is a HOST array.
 * snap_expectation_nd_f32: pdf [rows, size...] -> out [rows, n] = sum_cells index(cell) pdf(cell)
 * (:148-153; deterministic fixed-order sums).  argmax_nd (:140-145) = snap_argmax_rows_f32 over
 * the flattened grid + GridND.id_to_index on the host side. */
int snap_interpolate_nd_f32(const float* array, const int32_t* size, int32_t n, int32_t D,
                            const uint8_t* valid_array, const float* points, int64_t K,
                            float* values, uint8_t* valid, void* stream);
int snap_expectation_nd_f32(const float* pdf, int64_t rows, const int32_t* size, int32_t n,
                            float* out, void* stream);

/* Semantic-raster embedding (snap/models/semantic_raster_encoder.py:63-79).  rasters [M, N]
 * uint8 (bool); idx_road / idx_other: HOST arrays with the raster channels of the mutually
 * exclusive surfel-road classes and of the independent binary classes (<= 32 each);
 * table_road [nr, E], table_other [2*no, E]; out [M, (1 + no) * E]:
 * [table_road[argmax road bits] | table_other[j + bit_j] for j < no]  (the reference's row
 * indexing, :72-75).  E % 4 == 0.  snap_semantic_onehot_f32 writes the [M, KP] one-hot matrix
 * (column c < nr: road label c; column nr + 2j + b: class j has bit b; KP % 4 == 0) whose
 * transpose times d out (snap_conv2d_wgrad_f32) gives the table gradients deterministically. */
int snap_semantic_embed_f32(const uint8_t* rasters, int64_t M, int32_t N, const int32_t* idx_road,
                            int32_t nr, const int32_t* idx_other, int32_t no,
                            const float* table_road, const float* table_other, int32_t E,
                            float* out, void* stream);
int snap_semantic_onehot_f32(const uint8_t* rasters, int64_t M, int32_t N, const int32_t* idx_road,
                             int32_t nr, const int32_t* idx_other, int32_t no, float* onehot,
                             int32_t KP, void* stream);

/* ViT encoder pieces (BASELINE.json configs[4]; the reference itself has no ViT --
 * snap/models/image_encoder.py:103 -- so these follow the published ViT block).
 * snap_layer_norm_f32: y[m,:] = (x[m,:] - mean) * rsqrt(var + eps) * gamma + beta over the C
 * channels of each of the M rows (biased variance); C % 4 == 0, C <= 1024.
 * snap_attention_bf16_f32: multi-head self-attention softmax(scale * Q K^T) V on the bf16
 * matrix cores (f32 softmax statistics and accumulation).  qkv [B, N, 3, H, D] f32 (the fused
 * QKV projection's output), out [B, N, H*D] f32; D must be 64. */
int snap_layer_norm_f32(const float* x, const float* gamma, const float* beta, float* y,
                        int64_t M, int32_t C, float eps, void* stream);
int snap_attention_bf16_f32(const float* qkv, float* out, int32_t B, int32_t N, int32_t H,
                            int32_t D, float scale, void* stream);
/* The same two with the result written ONLY rounded (RNE) to bf16 (y_bf16 [M, C] / out_bf16 [B, N, H*D]): the
 * inference path's operands of the dense layers that follow (snap_conv2d_nhwc_ex_f32 with extras->x_presplit = 1,
 * w_split_parts = 1: both operands by LDS-DMA), which round them to that type anyway -- the same values. */
int snap_layer_norm_bf16out_f32(const float* x, const float* gamma, const float* beta, void* y_bf16,
                                int64_t M, int32_t C, float eps, void* stream);
int snap_attention_bf16out_f32(const float* qkv, void* out_bf16, int32_t B, int32_t N, int32_t H,
                               int32_t D, float scale, void* stream);
/* ... and with qkv itself in bf16 (the QKV projection's bf16-only output, y_half): K and V are the values the f32 form
 * rounds to, moved at half the bytes (the kernel is bound by the L2 traffic of the K / V panels); Q is scaled after
 * its rounding. */
int snap_attention_bf16io(const void* qkv_bf16, void* out_bf16, int32_t B, int32_t N, int32_t H, int32_t D,
                          float scale, void* stream);
/* Training path of the ViT pieces.  snap_attention_lse_bf16_f32 also returns lse [B, H, N], the
 * base-2 log-sum-exp of the scaled scores; snap_attention_bwd_bf16_f32 turns (qkv, out, dout,
 * lse) into dqkv (same layout as qkv; delta [B, H, N] is scratch) with two atomic-free kernels
 * on the bf16 matrix cores.  snap_layer_norm_bwd_f32: dx, dgamma, dbeta (fixed-order column
 * sums; workspace from snap_layer_norm_bwd_workspace_bytes).  snap_gelu[_bwd]_f32: tanh-form
 * GELU and its VJP over n (% 4 == 0) elements (the training path keeps the pre-activation). */
int snap_attention_lse_bf16_f32(const float* qkv, float* out, float* lse, int32_t B, int32_t N,
                                int32_t H, int32_t D, float scale, void* stream);
int snap_attention_bwd_bf16_f32(const float* qkv, const float* out, const float* dout,
                                const float* lse, float* delta, float* dqkv, int32_t B, int32_t N,
                                int32_t H, int32_t D, float scale, void* stream);
size_t snap_layer_norm_bwd_workspace_bytes(int64_t M, int32_t C);
int snap_layer_norm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx,
                            float* dgamma, float* dbeta, int64_t M, int32_t C, float eps,
                            void* workspace, size_t workspace_bytes, void* stream);
int snap_gelu_f32(const float* x, float* y, int64_t n, void* stream);
int snap_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* Scratch for split-K launches (small-M / deep-K layers that cannot fill 256 CUs with output
 * tiles: slices of K go to extra workgroups, partial tiles are summed in fixed order by a
 * second kernel that also applies the epilogue -- deterministic).  0 = the shape does not
 * split.  Without a workspace the launch simply does not split. */
size_t snap_conv2d_workspace_bytes(const SnapConvDesc* desc);
size_t snap_conv2d_gn_partial_bytes(const SnapConvDesc* desc);
int32_t snap_conv2d_tile_rows(const SnapConvDesc* desc);   /* row-tile height the launch uses */
/* ... of a launch on the split-operand engine with `split_parts` parts (SnapConvExtras.w_split_parts;
 * 0 = any other engine): the weights-stationary 1x1 kernel of the two-part engine emits its
 * GroupNorm partial sums per 32-row slab; a NEGATIVE value -S: S slabs per image, all of them live
 * (the weights-stationary 3x3 kernel: one slab per row-aligned tile) -- pass it on to
 * snap_group_norm_stats_from_partial_f32 as it is. */
int32_t snap_conv2d_tile_rows_ex(const SnapConvDesc* desc, int32_t split_parts);
size_t snap_conv2d_gn_partial_bytes_ex(const SnapConvDesc* desc, int32_t split_parts);
size_t snap_conv2d_splitk_gn_partial_bytes(const SnapConvDesc* desc);   /* see SnapConvExtras.gn_partial_rows */
/* Which body a split-bf16 launch (w_split_parts parts, no row lists) of this descriptor takes:
 * 0 = the tiled body, 1 = the row-stationary 1x1 kernel (activation tile in registers), 2 = the
 * weights-stationary 1x1 kernel (panel resident in LDS) -- conv_rs.hip: the bottleneck units' closing
 * / projection convolution, snap/models/resnet.py:112-132.  Same output bits whichever it is; a pure
 * function of the descriptor (desc->tile_hint carries the A/B mode).  For profiling labels and tests. */
int32_t snap_conv2d_stationary_kind(const SnapConvDesc* desc, int32_t parts);

/* mu / sc (/ rstd) [N, C] from a conv launch's gn_partial.  tile_rows =
 * snap_conv2d_tile_rows(desc of that launch); HW = Ho*Wo of its output. */
int snap_group_norm_stats_from_partial_f32(const float* partial, int32_t N, int32_t HW,
                                           int32_t C, int32_t groups, float eps,
                                           int32_t tile_rows, const float* gamma, float* mu,
                                           float* sc, float* rstd, void* stream);

/* Ascending list of the rows with mask != 0: index[0..count) (stable order,
 * deterministic), count written to *count (device).  index must hold M entries. */
size_t snap_compact_rows_workspace_bytes(int64_t M);
int snap_compact_rows_u8(const uint8_t* mask, int64_t M, int32_t* index, int32_t* count,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same list for the rows with lo <= mask[m] <= hi (classed masks: SnapLiftDesc.class_rows). */
int snap_compact_rows_range_u8(const uint8_t* mask, int64_t M, int32_t lo, int32_t hi,
                               int32_t* index, int32_t* count, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- fusion MLP + vertical max pooling in one kernel (mlp_pool.hip) --------------------
 * The two Dense layers of StreetViewEncoder.fusion_mlp (streetview_encoder.py:279-286,
 * layers.py:55-78: Dense(H) -> relu -> Dense(D)) over the rows listed in `rows`, followed by
 * VerticalPooling('max') (bev_mapper.py:78-88) over the Z levels of a column:
 *   plane[col, :] = max over listed rows m with rows[m] / Z == col of Dense1(relu(Dense0(x[rows[m]])))
 *   pvalid[col]   = the column has a listed row;  plane = 0 where it has none.
 * Neither the hidden activations nor the [.., Z, D] feature volume are written.  Arithmetic:
 * the split-bf16 engine with 2 parts ("bf16x3"); the plane equals, bit for bit, the one
 * snap_conv2d_nhwc_ex_f32 (w_split_parts = 2) x 2 + snap_fill_masked_rows_f32 +
 * snap_vertical_pool_f32 produce.
 *   x [*, x_stride] f32 (Cin <= x_stride, x_stride % 4 == 0); rows / row_count from
 *   snap_compact_rows_u8 (ascending); M = upper bound of the row count;
 *   w0_split = snap_conv2d_pack_weights_split_bf16(W0 [Cin, H], taps 1, parts 2);
 *   w1_split = the same packing of W1 [H, D];  H % 32 == 0, H <= 256; D % 4 == 0, D <= 128;
 *   relu_in: MLP.apply_input_activation;  plane [ncols, D], pvalid [ncols].
 *   x_split = 1: x holds the rows pre-split (SnapLiftDesc.out_split: [slab][hi | lo][16] bf16,
 *   x_stride still in floats); the A operand then travels global -> LDS by LDS-DMA.  Same
 *   results bit for bit; relu_in must be 0.  x_split = 5 (tuning / tests): the same with the GEMM0
 *   slabs one ahead (two LDS stages) instead of two ahead (three).  x_split = 3 (tuning / tests; H = 256 only, else as 1):
 *   the rows are taken 256 per workgroup (64 per wave, accumulators in AccVGPRs; the weights cross
 *   L2 -> LDS once per 256 rows) -- same bits, measured slower than the 128-row kernel. */
int snap_mlp2_pool_max_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                           const int32_t* rows, const int32_t* row_count,
                           const void* w0_split, size_t w0_bytes, const float* b0, int32_t H,
                           const void* w1_split, size_t w1_bytes, const float* b1, int32_t D,
                           int32_t relu_in, int32_t x_split, int32_t Z, int64_t ncols,
                           float* plane, uint8_t* pvalid, void* stream);
/* Two row classes into ONE plane: `rows` as above, and `rows_z` (may be NULL) whose rows are
 * exactly zero over the 16-channel slabs [zero_slab_lo, zero_slab_lo + zero_slabs) of the
 * first Dense layer's input and need not hold them in memory (single-observation rows of a
 * lift with SnapLiftDesc.class_rows: the variance slabs).  Those slabs are not read and not
 * multiplied for that class -- a product with +0 leaves an f32 accumulator unchanged, so the
 * plane equals the one-list result bit for bit; max is order-independent across the two. */
int snap_mlp2_pool_max_classes_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                                   const int32_t* rows, const int32_t* row_count,
                                   const int32_t* rows_z, const int32_t* row_count_z,
                                   int32_t zero_slab_lo, int32_t zero_slabs,
                                   const void* w0_split, size_t w0_bytes, const float* b0, int32_t H,
                                   const void* w1_split, size_t w1_bytes, const float* b1, int32_t D,
                                   int32_t relu_in, int32_t x_split, int32_t Z, int64_t ncols,
                                   float* plane, uint8_t* pvalid, void* stream);

/* The lift INSIDE the consumer (replaces k3-k6 for voxels with one visible observation:
 * streetview_encoder.py:69-105 gather, :141-178 pooling, :279-286 fusion MLP; bev_mapper.py:78-88 max):
 * `rows_g` lists voxels whose `pooled` row does not exist; tap_records [*, 8] u32
 * (snap_lift_pool_records_f32) hold their four-tap bilinear record, and the kernel gathers and blends
 * 16 channels of the four taps per GEMM0 slab while it stages the operand (mean = the observation,
 * variance slabs = 0: skipped, score slab from the record).  `rows` (several observations) are read
 * pre-split from x (SnapLiftDesc.out_split) as snap_mlp2_pool_max_classes_f32 does.  The plane equals
 * that entry point's bit for bit.  f_images [.., img_w, img_C] f32, f_bytes < 4 GB; Cin = 2 fd + 1,
 * feature_dim % 16 == 0; xcd_group: runs of that many 128-row tiles of rows_g per XCD (0: dispatch
 * order; results do not depend on it). */
int snap_mlp2_pool_max_gather_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                                  const int32_t* rows, const int32_t* row_count,
                                  const int32_t* rows_g, const int32_t* row_count_g,
                                  const float* f_images, int64_t f_bytes, int32_t img_w,
                                  int32_t img_C, int32_t feature_dim, const uint32_t* tap_records,
                                  int32_t xcd_group, const void* w0_split, size_t w0_bytes,
                                  const float* b0, int32_t H, const void* w1_split, size_t w1_bytes,
                                  const float* b1, int32_t D, int32_t Z, int64_t ncols, float* plane,
                                  uint8_t* pvalid, void* stream);

/* y[m, 0..C) = value for every row with mask[m] == 0 (the masked voxels of a volume
 * whose observed rows were written through rows_out).  C % 4 == 0. */
int snap_fill_masked_rows_f32(float* y, const uint8_t* mask, int64_t M, int32_t C,
                              float value, void* stream);

/* pad_to_multiple (image_encoder.py:32-39) and optional zero channels in one pass:
 * x [N, H, W, C] -> y [N, H + pad_h, W + pad_w, C + pad_c], zeros bottom / right / extra channels. */
int snap_pad_image_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad_h,
                       int32_t pad_w, int32_t pad_c, float* y, void* stream);

/* Voxel-centre query points (bev_mapper.py:162-196): out [B, XY, Z, 3] = (xy[b, c, :], z[b, k]);
 * xy [XY, 2] (xy_batched = 0) or [B, XY, 2]; z [B, Z] (level heights per scene). */
int snap_voxel_points_f32(const float* xy, int32_t xy_batched, const float* z, int32_t B, int32_t XY,
                          int32_t Z, float* out, void* stream);

/* StdConv weight standardisation over (H,W,I) per output channel, eps inside the
 * sqrt (resnet.py:34-41,73-79).  w,out: [K, Cout]. */
int snap_weight_standardize_f32(const float* w, float* out, int32_t K,
                                int32_t Cout, float eps, void* stream);

/* All StdConv kernels of an encoder in one launch (training standardises ~53 kernels per
 * encoder every step; one launch per kernel is pure launch latency).  `items` is a DEVICE
 * array; item i owns workgroups [block_begin, block_begin + ceil(Cout / 32)) of the forward
 * and of the backward launch; items sorted by block_begin; total_blocks = sum.  Forward: out = standardise(w) (dws unused).
 * Backward (snap_weight_standardize_bwd_multi_f32): out = d w given dws = d standardise(w). */
typedef struct SnapWstdItem {
  const float* w;
  const float* dws;
  float* out;
  int32_t K, Cout, block_begin, reserved;
} SnapWstdItem;
int snap_weight_standardize_multi_f32(const SnapWstdItem* items, int32_t n_items,
                                      int32_t total_blocks, float eps, void* stream);
int snap_weight_standardize_bwd_multi_f32(const SnapWstdItem* items, int32_t n_items,
                                          int32_t total_blocks, float eps, void* stream);

/* GroupNorm statistics (resnet.py:46-60): two-pass mean / mean((x-mean)^2) over
 * (H,W,C/G) per (image, group).  relu_first != 0 computes them on relu(x)
 * (FPN order, image_encoder.py:80-83).  Outputs mu[N,C], sc[N,C] = rstd*gamma[c].
 * workspace: snap_group_norm_stats_workspace_bytes(N, HW, C, groups) bytes. */
size_t snap_group_norm_stats_workspace_bytes(int32_t N, int32_t HW, int32_t C,
                                             int32_t groups);
int snap_group_norm_stats_f32(const float* x, int32_t N, int32_t HW, int32_t C,
                              int32_t C_stride, int32_t groups, float eps,
                              int32_t relu_first, const float* gamma, float* mu,
                              float* sc, float* rstd /* optional [N,C] */,
                              void* workspace, size_t workspace_bytes,
                              void* stream);

/* Stand-alone GroupNorm apply (+optional ReLU before/after); used by tests and
 * for returning normalised tensors.  mode: SNAP_PRO_GN_RELU / SNAP_PRO_RELU_GN. */
int snap_group_norm_apply_f32(const float* x, float* y, int32_t N, int32_t HW,
                              int32_t C, const float* mu, const float* sc,
                              const float* beta, int32_t mode, void* stream);

/* 3x3 / stride 2 / pad 1 max-pool with -inf padding (resnet.py:99). NHWC. */
int snap_max_pool_3x3s2_f32(const float* x, float* y, int32_t N, int32_t H,
                            int32_t W, int32_t C, void* stream);

/* ------------------------------------------------------------------------- *
 * Lift: voxel -> view projection, top-K view selection, bilinear gather, depth
 * score interpolation and softmax-weighted multi-view pooling, fused.
 *   replaces snap/models/streetview_encoder.py:42-178 (k1-k5 of SURVEY 2.2) with
 *   snap/utils/geometry.py:52-69,198-221,260-280.
 * ------------------------------------------------------------------------- */
typedef struct SnapLiftDesc {
  int32_t B, V, h, w, C;       /* f_images [B,V,h,w,C], C = feature_dim + score bins */
  int32_t feature_dim;         /* 128 */
  int32_t num_bins;            /* 32  */
  int32_t N;                   /* voxels per scene */
  int32_t K;                   /* 0 => use all V views (interpolate_views_all),
                                  else top-K selection (V > K)                     */
  int32_t fisheye;             /* 1: FisheyeCamera, 0: pinhole Camera              */
  int32_t out_stride;          /* row stride (floats) of `pooled`, >= 2*fd+1, %4==0 */
  float depth_min, depth_max;  /* depth_min_max                                    */
  float max_view_distance;     /* < 0: disabled                                    */
  /* fusion options of pool_multiview_features (streetview_encoder.py:141-178); the reference
   * default is weighted = 1, use_variance = 1, add_minmax = 0 */
  int32_t weighted;            /* do_weighted_fusion: depth-score softmax weights + the
                                  score_max channel; 0: plain mean / variance (scores = None),
                                  f_images carries no score bins (C = feature_dim)  */
  int32_t use_variance;        /* fusion_use_variance                              */
  int32_t add_minmax;          /* fusion_add_minmax: + max, min over the valid views */
  /* traversal hint (results do not depend on it): the N points of a scene are the voxel
   * centres of an [N / (grid_y * grid_z), grid_y, grid_z] grid, level fastest
   * (bev_mapper.py:162-196); the kernel then walks 8 x 8 blocks of columns per XCD so that a
   * block's image taps stay in that XCD's L2.  0, 0 = unstructured points (query frustum). */
  int32_t grid_y, grid_z;
  /* 1: rows of `pooled` whose voxel no view sees are NOT written (their `valid` byte is 0;
   * the default options' batched kernel only) -- for consumers that read the rows of valid
   * voxels only (snap_mlp2_pool_max_f32, row-indexed conv launches).  0: zeros, as the
   * reference's pool_multiview_features returns. */
  int32_t valid_rows_only;
  /* 1: `pooled` rows are written pre-split for the split-bf16 engines instead of as f32:
   * row = [ceil(channels / 16) slabs][hi | lo][16] bf16 (64 B per slab; hi = bf16(v) RNE,
   * lo = bf16(v - hi)), out_stride (in floats) >= 16 * slabs -- what snap_mlp2_pool_max_f32
   * takes with x_split = 1.  Default fusion options, <= 4 selected views, feature_dim % 8 == 0. */
  int32_t out_split;
  /* 1 (with out_split; feature_dim % 16 == 0): rows are CLASSED by their number of visible
   * observations -- valid[] = 0 invalid, 1 one observation, 2 several -- and a one-observation
   * row does not write its variance slabs (slabs feature_dim/16 .. 2 feature_dim/16 - 1: the
   * weighted variance of a single observation is exactly 0; 72 % of the observed voxels of a
   * four-view map and every voxel of the query).  The consumer lists the two classes
   * separately (snap_compact_rows_range_u8) and skips those slabs for the first
   * (snap_mlp2_pool_max_classes_f32): 47 % fewer bytes written and re-read for such a row, same
   * plane bits (products with +0 leave an accumulator unchanged). */
  int32_t class_rows;
  /* A/B switches for tests and tools (0 = defaults).  Bit 0: snap_lift_pool_bwd_det_f32 takes the
   * half-wave-per-voxel record producer also where the batched one applies. */
  int32_t tune_flags;
} SnapLiftDesc;

/* cam: [B,V,11] = wh(2) f(2) c(2) k_radial(3) max_fov(1) tan(max_fov/2)(1) ALREADY scaled to
 * the feature-map resolution (the last entry is read by the fisheye path instead of
 * evaluating tanf per voxel and view: geometry.py:262 `radius < tan(0.5 * max_fov)`); Rt: [B,V,12] = R row-major (9) then t (3) of
 * T_view2scene; points: [B,N,3].
 * pooled: [B,N,out_stride] = mean(fd) | var(fd)? | max(fd), min(fd)? | score_max(1)? | zero pad
 * (default options: mean | var | score_max);  valid: [B,N] uint8. */
int snap_lift_pool_f32(const SnapLiftDesc* desc, const float* f_images,
                       const float* cam, const float* Rt, const float* points,
                       float* pooled, uint8_t* valid, void* stream);

/* The same with TAP RECORDS (class_rows, out_split and valid_rows_only set): a voxel with ONE visible
 * observation writes no row into `pooled` but tap_records[voxel][8] (u32) = byte offset of tap
 * (i0, j0) channel 0 in f_images | bit 8: i1 != i0, bit 9: j1 != j0 (the clamped upper taps), depth
 * bins | wi1 | wj1 (f32 bits) | depth score (f32 bits) | 0 0 0 -- streetview_encoder.py:93-124 for
 * that observation; its mean is the bilinear blend (softmax weight exactly 1), its variance 0.
 * snap_mlp2_pool_max_gather_f32 consumes the records.  Voxels with several observations write
 * their rows as above; valid[] carries the classes 0 / 1 / 2. */
int snap_lift_pool_records_f32(const SnapLiftDesc* desc, const float* f_images,
                               const float* cam, const float* Rt, const float* points,
                               float* pooled, uint8_t* valid, uint32_t* tap_records, void* stream);

/* depth_mlp fusion (streetview_encoder.py:263-267; do_weighted_fusion = False, so desc->weighted
 * = 0 and C = feature_dim): the lift in two passes around a per-observation MLP.
 *   observations: obs [B, N, S, fd + 4] = bilinear features | log10(clip(depth, 0.1, 100)) |
 *     unit ray in the camera frame (zero where the view does not see the point), S = K (top-K
 *     order) or V (view order); obs_feat [B, N, S, fd] = the features alone (the residual the
 *     caller adds the MLP output to); valid [B, N] as snap_lift_pool_f32.
 *   pool: pool_multiview_features(obs_feat', visible, scores = None, add_minmax, use_variance)
 *     of the corrected observations obs_feat' [B, N, S, fd] -> pooled [B, N, out_stride], valid. */
int snap_lift_observations_f32(const SnapLiftDesc* desc, const float* f_images, const float* cam,
                               const float* Rt, const float* points, float* obs, float* obs_feat,
                               uint8_t* valid, void* stream);
int snap_lift_pool_observations_f32(const SnapLiftDesc* desc, const float* cam, const float* Rt,
                                    const float* points, const float* obs_feat, float* pooled,
                                    uint8_t* valid, void* stream);

/* Debug/parity variant of k1: p2d[B,N,V,2] (ij), vis[B,N,V], depth[B,N,V]. */
int snap_project_points_f32(int32_t B, int32_t V, int32_t N, int32_t fisheye,
                            const float* cam, const float* Rt,
                            const float* points, float* p2d, uint8_t* vis,
                            float* depth, void* stream);

/* ------------------------------------------------------------------------- *
 * BEV: vertical pooling, modality fusion and matching head.
 *   replaces snap/models/bev_mapper.py:56-88 (k7), :225-252 (k8), :284-291 +
 *   snap/models/layers.py:45-52 (k9).
 * ------------------------------------------------------------------------- */
enum { SNAP_POOL_MAX = 0, SNAP_POOL_SUM = 1, SNAP_POOL_MEAN = 2 };

/* vol [M, Z, D] + vvalid [M, Z] -> plane [M, D], pvalid [M]. */
int snap_vertical_pool_f32(const float* vol, const uint8_t* vvalid, float* plane,
                           uint8_t* pvalid, int64_t M, int32_t Z, int32_t D,
                           int32_t pooling, void* stream);
/* The max pooling above (Z <= 64, D <= 128) that also records, per (column, channel), the level of the
 * maximum -- the first of equal ones, found with the comparisons of the VJP (`>` then `==` over the
 * valid levels in ascending order, a NaN never wins) -- and how many levels hold it (saturating at
 * 255): argz [M, D], ties [M, D] bytes (255 / 0 for a column without a valid level).  What
 * snap_vertical_pool_max_bwd_arg_f32 reads instead of the volume (VerticalPooling('max') in a
 * training step: snap/models/bev_mapper.py:56-88). */
int snap_vertical_pool_max_arg_f32(const float* vol, const uint8_t* vvalid, float* plane,
                                   uint8_t* pvalid, uint8_t* argz, uint8_t* ties, int64_t M,
                                   int32_t Z, int32_t D, void* stream);

/* Fuse `num_planes` modality planes (planes[i] [M,D], valids[i] [M] or NULL=all
 * valid) with masked max/sum/mean, then matching head: Dense(D->Dm)+bias, L2
 * normalise (eps), mask.  fused [M,D], fvalid [M], matching [M,Dm] (may be NULL
 * with Wm NULL). */
int snap_plane_fuse_match_f32(const float* const* planes,
                              const uint8_t* const* valids, int32_t num_planes,
                              int64_t M, int32_t D, int32_t pooling, float* fused,
                              uint8_t* fvalid, const float* Wm, const float* bm,
                              int32_t Dm, int32_t normalize, float eps,
                              float* matching, void* stream);

/* ------------------------------------------------------------------------- *
 * Pose: point-vs-map similarity, sampling, scoring, refinement.
 *   replaces snap/models/bev_localizer.py:157-173 (k10),
 *   snap/models/pose_estimation.py:126-165 (k11), :49-82 (k12), :168-205 (k13).
 * ------------------------------------------------------------------------- */
#define SNAP_SIM_CHUNK 64  /* cells per (max, sum) softmax chunk == one wave64 */

/* sim[B,Nq,XY] = relu(fq . fm) * scale / num_valid[b]  (relu iff clip_negative);
 * chunk_stats[B,Nq,ceil(XY/64),2] = per-chunk (max, sum exp(x - max)) of the
 * UN-normalised x = relu(.)*scale: enough to evaluate / sample
 * prob = softmax_XY(x) / num_valid without a second [B,Nq,XY] tensor.
 * Optional (may both be NULL): rowstats[B,Nq,2] = (row max, row sum) and
 * prob[B,Nq,XY] = the full prob_points tensor (needs rowstats).
 * num_valid: [B] float, already clipped to >= 1 (bev_localizer.py:171). */
int snap_sim_softmax_f32(const float* fq, const float* fm, int32_t B, int32_t Nq,
                         int32_t XY, int32_t Dm, float scale,
                         int32_t clip_negative, const float* num_valid,
                         float* sim, float* chunk_stats, float* prob,
                         float* rowstats, void* stream);

/* Bytes of the optional rowstats [B, Nq, 2] buffer of the similarity entry points (caller-owned). */
size_t snap_sim_rowstats_bytes(int32_t B, int32_t Nq);

/* add_confidence_query (bev_localizer.py:165-168): the 1 / num_valid normalisation of sim (and
 * prob) is replaced by per-point weights row_weight[B, Nq] = layers.masked_softmax(bev_confidence
 * of the query points, valid points).  row_weight == NULL is snap_sim_softmax_f32. */
int snap_sim_softmax_weighted_f32(const float* fq, const float* fm, int32_t B, int32_t Nq,
                                  int32_t XY, int32_t Dm, float scale, int32_t clip_negative,
                                  const float* num_valid, const float* row_weight, float* sim,
                                  float* chunk_stats, float* prob, float* rowstats, void* stream);
/* The same sim / chunk_stats with the Nq x XY x Dm contraction on the bf16 matrix cores at f32
 * grade: fq / fm are split into `parts` bf16 parts per element (3: six products per MAC, ~2^-24
 * per product -- the arithmetic of the conv engine's 'bf16x6'; 2: three products, ~2^-17) into
 * `workspace` first.  Dm in {16, 32, 64}.  (prob / rowstats: use snap_sim_softmax_weighted_f32.)
 * XY % 256 == 0 takes a leaner kernel with the same bits; clip_negative bit 1 (value 2) pins the
 * general kernel (tests compare the two). */
size_t snap_sim_split_workspace_bytes(int32_t B, int32_t Nq, int32_t XY, int32_t Dm, int32_t parts);
int snap_sim_softmax_split_f32(const float* fq, const float* fm, int32_t B, int32_t Nq, int32_t XY,
                               int32_t Dm, float scale, int32_t clip_negative,
                               const float* num_valid, const float* row_weight, int32_t parts,
                               float* sim, float* chunk_stats, void* workspace,
                               size_t workspace_bytes, void* stream);
/* layers.masked_softmax over the last axis (snap/models/layers.py:38-43: an all-false mask acts
 * as all-true) of x [B, N] + its inclusive CDF (the sampler's distribution over query points). */
int snap_masked_softmax_rows_f32(const float* x, const uint8_t* mask, int32_t B, int32_t N,
                                 float* weights, float* cdf, void* stream);
/* bev_confidence = where(valid, log_sigmoid(features . w + bias), 0)  (Dense(1) head of the BEV
 * plane, bev_mapper.py:154-157,292-295).  features [M, D], D % 4 == 0; valid may be NULL. */
int snap_confidence_head_f32(const float* features, const uint8_t* valid, const float* w,
                             const float* bias /* DEVICE scalar */, int64_t M, int32_t D, float* out,
                             void* stream);
/* Its VJP: d features [M, D] = ds w; prod [M, D] = ds f and dsv [M, 4] = (ds, 0, 0, 0) are the
 * per-row terms whose fixed-order column sums (snap_colsum_f32) are d w and d bias;
 * ds = dconf * sigmoid(-(f . w + bias)) * [valid]. */
int snap_confidence_head_bwd_f32(const float* features, const uint8_t* valid, const float* w,
                                 const float* bias, const float* dconf, int64_t M, int32_t D,
                                 float* dfeatures, float* prod, float* dsv, void* stream);
/* add_confidence_query (bev_localizer.py:165-168), training: rowdot[b, n] = sum_cells dsim * sim
 * (the row's share of the temperature gradient; / weight = the gradient of the point weight), then
 * dsim = dsim * [sim > 0] * row_coef[b, n] in place; and the VJP of layers.masked_softmax
 * (layers.py:38-43): dx = w * (dw - sum_n w dw). */
int snap_sim_bwd_prepare_rows_f32(float* dsim, const float* sim, int32_t B, int32_t Nq, int32_t XY,
                                  int32_t clip_negative, const float* row_coef, float* rowdot,
                                  void* stream);
int snap_masked_softmax_rows_bwd_f32(const float* weights, const float* dweights, int32_t B,
                                     int32_t N, float* dx, void* stream);

/* Draw S correspondences per scene ~ prob_points (iid categorical; counter-based
 * Philox4x32-10 keyed by (seed, b, s)).  corr[B,S,3] = (n, i, j).
 * If uniforms != NULL ([B,S,2] in [0,1)), they replace the Philox draws (tests). */
int snap_ransac_sample_f32(const float* fq, const float* fm,
                           const float* chunk_stats, int32_t B, int32_t Nq,
                           int32_t X, int32_t Y, int32_t Dm, float scale,
                           int32_t clip_negative, int32_t S, uint64_t seed,
                           const float* uniforms, int32_t* corr, void* stream);

/* Same samples, faster: with a workspace (snap_ransac_sample_workspace_bytes) the per-row
 * prefix of the chunk masses is built once per row instead of once per sample (~34 samples
 * share a row at the default sizes).  Bit-identical output. */
size_t snap_ransac_sample_workspace_bytes(int32_t B, int32_t Nq);
/* ... with sim [B, Nq, X*Y] (the tensor snap_sim_softmax*_f32 wrote) and row_unscale [B, Nq]
 * (num_valid[b], or 1 / row_weight[b, n]: x = sim * row_unscale) the selected chunk's 64 scores
 * are READ (256 contiguous bytes) instead of re-evaluated from 8 KB of map features; both NULL =
 * re-evaluate.  Same distribution; the scores agree to one rounding. */
int snap_ransac_sample_sim_f32(const float* fq, const float* fm, const float* chunk_stats,
                               const float* row_cdf, const float* sim, const float* row_unscale,
                               int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t Dm, float scale,
                               int32_t clip_negative, int32_t S, uint64_t seed,
                               const float* uniforms, int32_t* corr, void* workspace,
                               size_t workspace_bytes, void* stream);
/* ... with row_cdf [B, Nq] (inclusive, from snap_masked_softmax_rows_f32) the query point of a
 * sample is drawn from that distribution instead of uniformly (confidence-weighted prob_points). */
int snap_ransac_sample_rows_f32(const float* fq, const float* fm, const float* chunk_stats,
                                const float* row_cdf, int32_t B, int32_t Nq, int32_t X, int32_t Y,
                                int32_t Dm, float scale, int32_t clip_negative, int32_t S,
                                uint64_t seed, const float* uniforms, int32_t* corr,
                                void* workspace, size_t workspace_bytes, void* stream);
int snap_ransac_sample_ws_f32(const float* fq, const float* fm, const float* chunk_stats,
                              int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t Dm,
                              float scale, int32_t clip_negative, int32_t S, uint64_t seed,
                              const float* uniforms, int32_t* corr, void* workspace,
                              size_t workspace_bytes, void* stream);

/* snap_pose_score_f32 for pose sets CLUSTERED around one centre pose per scene -- the 41^3
 * refinement lattice of grid_refinement (pose_estimation.py:168-205): every pose of scene b maps every
 * query point to within `radius_cells` cells of where centers[b] (angle, tx, ty) maps it (the caller
 * guarantees it: |t - t_c| / cell + max |q| * |angle - angle_c| / cell, rounded up, + 1).  A point's
 * score plane is then read as ONE window of <= (2 r + 3) x (2 r + 6) cells instead of whole, once per
 * pose chunk (256 x 256 maps: 27 KB instead of 262 KB per point).  Same arithmetic per sample and the
 * same order of every sum as snap_pose_score_f32 (mask_oob = 0): the scores are the same bits.
 * snap_pose_score_window_supported: 1 when the window of that radius fits the kernel's LDS buffers
 * (else call snap_pose_score_f32).  A pose outside the promised radius reads a clamped cell of the
 * window (a wrong score, never out of bounds). */
size_t snap_pose_score_window_workspace_bytes(int32_t B, int32_t Nq, int32_t P, int32_t X, int32_t Y);
int32_t snap_pose_score_window_supported(int32_t X, int32_t Y, int32_t radius_cells);
int snap_pose_score_window_f32(const float* sim, const float* poses, const float* centers,
                               int32_t radius_cells, const float* q_xy, const uint8_t* valid_q,
                               int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t P, float cell_size,
                               float* scores, void* workspace, size_t workspace_bytes, void* stream);

/* corr[B,P*retries*2,3] -> poses[B,P,3] = (angle, tx, ty) of map_t_query:
 * most distance-consistent retry, 2-point Kabsch (pose_estimation.py:146-165). */
int snap_poses_from_corr_f32(const int32_t* corr, const float* q_xy, int32_t B,
                             int32_t Nq, int32_t P, int32_t retries,
                             float cell_size, float* poses, void* stream);

/* scores[B,P] = sum_n valid_q[b,n] * bilinear(sim[b,n], (R(theta) q_xy[n] + t)/cell)
 * (pose_estimation.py:63-82).  poses[B,P,3]; map_valid [B,X,Y] only read when
 * mask_oob != 0.  workspace: snap_pose_score_workspace_bytes(B,Nq,P,X,Y). */
size_t snap_pose_score_workspace_bytes(int32_t B, int32_t Nq, int32_t P,
                                       int32_t X, int32_t Y);
int snap_pose_score_f32(const float* sim, const float* poses, const float* q_xy,
                        const uint8_t* valid_q, const uint8_t* map_valid,
                        int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t P,
                        float cell_size, int32_t mask_oob, float* scores,
                        void* workspace, size_t workspace_bytes, void* stream);

/* out[B,nr*np*np,3] = init[b] o offset(ir,ix,iy)  (pose_estimation.py:178-193).
 * offs_r[nr] (radians), offs_p[np] (metres). */
int snap_refine_lattice_f32(const float* init, const float* offs_r,
                            const float* offs_p, int32_t B, int32_t nr,
                            int32_t np_, float* out, void* stream);

/* Row-wise argmax with first-index tie-break: idx[B] over scores[B, start:P]. */
int snap_argmax_rows_f32(const float* scores, int32_t B, int32_t P, int32_t start,
                         int32_t* idx, void* stream);

/* ------------------------------------------------------------------------- *
 * Exhaustive voting: rotated templates + direct correlation.
 *   replaces snap/models/pose_exhaustive_voting.py:37-69 (k14), :72-104 (k15).
 * ------------------------------------------------------------------------- */
/* feat[H,W,D], valid[H,W] (H==W), tfm[R/4,4] = (cos, sin, tx, ty) of
 * templates_t_grid for the first quadrant of rotations ->
 * templates[R,H,W,D], tvalid[R,H,W]; also the engine-layout copies
 * tw[H,W,D,R] (HWIO; R % 4 == 0) and cw[H,W,1,R] = 180-degree-rotated tvalid as
 * float (the reference's un-flipped mask convolution,
 * pose_exhaustive_voting.py:97-99); tcount[R] = sum(tvalid[r]). */
int snap_rotate_templates_f32(const float* feat, const uint8_t* valid,
                              const float* tfm, int32_t H, int32_t W, int32_t D,
                              int32_t R, float cell_size, float* templates,
                              uint8_t* tvalid, float* tw, float* cw,
                              float* tcount, void* stream);

/* map[H,W,D], mvalid[H,W] -> map_pad[3H-2,3W-2,D] (edge), mvalid_pad[3H-2,3W-2]
 * (zero padded, as float). */
/* Shift-stacked filter bank for the exhaustive correlation: tw [H, W, D, R] (HWIO) ->
 * tws [H+S-1, W+S-1, D, R*S*S], tws[i', j', d, (r*S + sa)*S + sb] = tw[i'-sa, j'-sb, d, r]
 * (zero outside).  A stride-S snap_conv2d over the padded map with tws computes, at output
 * pixel (a4, b4) and channel (r, sa, sb), exactly the direct-form output (r, S a4 + sa,
 * S b4 + sb) of pose_exhaustive_voting.py:86-91 -- same products, full 64-wide GEMM tiles
 * instead of R = 36 columns. */
int snap_stack_templates_f32(const float* tw, float* tws, int32_t H, int32_t W, int32_t D,
                             int32_t R, int32_t S, void* stream);
/* The same bank from the [R, H, W, D] template tensor of snap_rotate_templates_f32 (whose `tw`
 * output may then be NULL: the r-fastest HWIO copy is only needed by the unstacked path). */
int snap_stack_templates_rhwd_f32(const float* templates, float* tws, int32_t H, int32_t W,
                                  int32_t D, int32_t R, int32_t S, void* stream);
/* ... and written DIRECTLY as the split engine's two-part weight image of that bank (what
 * snap_conv2d_pack_weights_split_bf16(tws, taps = (H+S-1)(W+S-1), Cin = D, Cout = R S^2, parts = 2)
 * would produce, bit for bit): the f32 bank is never materialised.  out:
 * snap_conv2d_packed_weights_split_bytes(taps, D, R S^2, 2) bytes, 16-byte aligned; D % 4 == 0. */
int snap_pack_stacked_templates_split_bf16(const float* templates, int32_t H, int32_t W, int32_t D,
                                           int32_t R, int32_t S, void* out, size_t out_bytes,
                                           void* stream);
int snap_pad_map_f32(const float* map, const uint8_t* mvalid, int32_t H, int32_t W,
                     int32_t D, float* map_pad, float* mvalid_pad, void* stream);

/* Frequency-domain template matching (snap/models/pose_exhaustive_voting.py:72-104, padded mode) in
 * ONE call: templates[R,H,W,D] (zero where tvalid is 0), tvalid[R,H,W], map[Hm,Wm,D], mvalid[Hm,Wm],
 * tcount[R] = q_valid.sum((-1,-2)) -> scores[R, 3Hm-1-H, 3Wm-1-W]:
 *   scores[r,a,b] = sum_ijd templates[r,i,j,d] * edge_pad(map)[a+i, b+j, d]            (:82-91)
 *   -inf where the overlap count (the 180-degree rotated template mask correlated with the
 *   zero-padded map mask, :93-101) is <= overlap_threshold (= min_overlap * H * W) if use_overlap,
 *   then / tcount[r]                                                                    (:103).
 * The correlation is evaluated as FFT products (channel pairs packed as complex numbers, mixed-radix
 * 2/3/4 transforms of 2^a or 3*2^a points per axis in LDS, f32): ~1e-6 relative to the largest
 * score instead of the direct form's summation order -- snap_conv2d_* on the stacked template bank
 * remains the direct form (and the checker).  The overlap count is rounded to the nearest integer
 * (operands are 0/1): the -inf mask is the direct form's.
 * workspace: snap_voting_fft_workspace_bytes() bytes, 256-byte aligned; 0 = unsupported geometry
 * (3*max(Hm,Wm)-2 > 1024 points per axis), for which snap_voting_fft_f32 returns
 * SNAP_ERR_UNSUPPORTED.  templates / map 8-byte aligned. */
size_t snap_voting_fft_workspace_bytes(int32_t R, int32_t H, int32_t W, int32_t D, int32_t Hm,
                                       int32_t Wm);
int snap_voting_fft_f32(const float* templates, const uint8_t* tvalid, const float* map,
                        const uint8_t* mvalid, const float* tcount, int32_t R, int32_t H, int32_t W,
                        int32_t D, int32_t Hm, int32_t Wm, float overlap_threshold,
                        int32_t use_overlap, void* workspace, size_t workspace_bytes, float* scores,
                        void* stream);

/* exhaustive_pose_voting (snap/models/pose_exhaustive_voting.py:107-124) in one call: the query plane
 * feat[H,H,D] (already multiplied by its confidence), valid[H,H], tfm[R/4,4] = (cos, sin, tx, ty) of
 * templates_t_grid for the first quadrant of rotations (:44-50), cell_size; map[Hm,Wm,D],
 * mvalid[Hm,Wm] -> scores[R, 3Hm-1-H, 3Wm-1-H] with min_overlap (0.05 in the reference).
 * sample_query_templates (:37-69) runs inside the first transform: template rows are interpolated on
 * the fly with the arithmetic of snap_rotate_templates_f32 (bit for bit), rotations R/4..R-1 are rot90
 * index maps; the [R,H,H,D] template tensor is never written.  R % 4 == 0, square query plane.
 * workspace: snap_voting_fft_workspace_bytes(R, H, H, D, Hm, Wm). */
int snap_voting_fft_rotated_f32(const float* feat, const uint8_t* valid, const float* tfm,
                                float cell_size, const float* map, const uint8_t* mvalid, int32_t R,
                                int32_t H, int32_t D, int32_t Hm, int32_t Wm, float min_overlap,
                                void* workspace, size_t workspace_bytes, float* scores, void* stream);

/* raw[Ho,Wo,Rp], cnt[Ho,Wo,Rp] (engine outputs) -> scores[R,Ho,Wo]:
 * -inf where cnt <= min_overlap*H*W (if min_overlap >= 0), then / tcount[r]. */
int snap_template_finalize_f32(const float* raw, const float* cnt,
                               const float* tcount, int32_t Ho, int32_t Wo,
                               int32_t R, int32_t Rp, float overlap_threshold,
                               int32_t use_overlap, float* scores, void* stream);

/* Confidence-weighted vertical pooling: the 'softmax' / 'weighted' modes of VerticalPooling
 * (bev_mapper.py:63-78).  score_z = vol[m,z,:] . w + bias[0] (log_sigmoid of it when
 * log_sigmoid_scores != 0, i.e. 'weighted'); weights = softmax over the valid levels (all
 * levels if none is valid, then zeroed), plane = sum_z weights_z vol[m,z,:], zero where no
 * level is valid.  scores / weights: [M, Z] (pred['scores'], pred['weights']).  Z <= 64,
 * D <= 128.  Backward: d vol from d plane (scores / weights outputs carry no gradient here),
 * plus per-workgroup partial rows [rows, D + 4] of (d w | d bias at column D) whose column
 * sums (snap_colsum_f32) are d w and d bias; rows = snap_vertical_pool_conf_bwd_partial_rows(M). */
int snap_vertical_pool_conf_f32(const float* vol, const uint8_t* vvalid, const float* w,
                                const float* bias, int64_t M, int32_t Z, int32_t D,
                                int32_t log_sigmoid_scores, float* plane, uint8_t* pvalid,
                                float* scores, float* weights, void* stream);
size_t snap_vertical_pool_conf_bwd_partial_rows(int64_t M);
int snap_vertical_pool_conf_bwd_f32(const float* vol, const uint8_t* vvalid, const float* w,
                                    const float* bias, const float* weights, const float* dplane,
                                    int64_t M, int32_t Z, int32_t D, int32_t log_sigmoid_scores,
                                    float* dvol, float* dw_partial, void* stream);

/* ------------------------------------------------------------------------- *
 * Training path: vector-Jacobian products of the kernels above.
 * The reference obtains these from jax.grad over the same call sites
 * (snap/trainer.py:223-234).  Data-gradients of convs reuse snap_conv2d_nhwc_f32
 * with the rotated / transposed kernel (host side, snap_amd/autograd.py).
 * ------------------------------------------------------------------------- */
/* dw[KH*KW*Cin, Cout] (+)= im2col(prologue(x))^T dy   on f32 MFMA; dy [N,Ho,Wo,Cout]. */
size_t snap_conv2d_wgrad_workspace_bytes(const SnapConvDesc* desc);
/* Per-call A/B switches (tools, tests) of the half-precision engines' plans, OR-ed into
 * desc->tile_hint of a snap_conv2d_wgrad_* call (no process-wide state):
 *   SNAP_WGRAD_NO_WIDE   flat kernel gradients (1 x 1, Cin >= 192, >= 65 536 rows) stay on the per-tap
 *                        128 x 128 tiles instead of the 512-thread 256-wide tiles;
 *   SNAP_WGRAD_NO_FUSED3 3 x 3 / stride 1 / pad 1 kernel gradients with a half-precision dy stay on the
 *                        per-tap kernel instead of the fused-tap kernel (all nine taps per workgroup,
 *                        patch-ordered reduction).
 * Same rounded operands, another summation order.  Default (0): both plans where they apply. */
enum { SNAP_WGRAD_NO_WIDE = 1 << 8, SNAP_WGRAD_NO_FUSED3 = 1 << 9 };
int snap_conv2d_wgrad_f32(const SnapConvDesc* desc, const float* x, const float* dy,
                          float* dw, const float* gn_mu, const float* gn_sc,
                          const float* gn_beta, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Row-list variant for the masked fusion MLP (flat [1,1,M,C] Dense layout only): reduction
 * row m reads x row rows_z[m] and dy row rows_dy[m]; only the first *row_count (device
 * scalar) rows exist.  Any of the three may be NULL. */
int snap_conv2d_wgrad_rows_f32(const SnapConvDesc* desc, const float* x, const float* dy,
                               float* dw, const float* gn_mu, const float* gn_sc,
                               const float* gn_beta, int32_t accumulate, void* workspace,
                               size_t workspace_bytes, const int32_t* rows_z,
                               const int32_t* rows_dy, const int32_t* row_count, void* stream);
/* The same with a choice of arithmetic: SNAP_MATH_F32 (exact f32 matrix cores) or
 * SNAP_MATH_BF16 (both operands rounded to bf16 after the f32 prologue, f32 accumulate: the
 * training-precision engine, see snap_conv2d_pack_weights_bf16).  Shapes the bf16 engine does
 * not carry (Cin < 4, unaligned channel rows) run in f32. */
#define SNAP_MATH_F32 0
#define SNAP_MATH_BF16 1
#define SNAP_MATH_F16 2   /* operands rounded to IEEE half (the reference's float16 train config) */
int snap_conv2d_wgrad_ex_f32(const SnapConvDesc* desc, const float* x, const float* dy,
                             float* dw, const float* gn_mu, const float* gn_sc,
                             const float* gn_beta, int32_t accumulate, void* workspace,
                             size_t workspace_bytes, const int32_t* rows_z,
                             const int32_t* rows_dy, const int32_t* row_count, int32_t math,
                             void* stream);

/* ... with ONE of the operands already in the engine's 2-byte element type (math BF16 / F16):
 * x_is_half -- x [rows, Cin_stride] (prologue NONE), dy_is_half -- dy [rows, Cout_stride]; the other
 * operand stays f32.  The hidden activations and inter-layer gradients of the masked MLP, written in
 * that type by SnapConvExtras.y_half / snap_epilogue_bwd_colsum_half: half the bytes, no rounding
 * work, same bits as rounding an f32 copy in the loop. */
int snap_conv2d_wgrad_half_f32(const SnapConvDesc* desc, const void* x, const void* dy, float* dw,
                               const float* gn_mu, const float* gn_sc, const float* gn_beta,
                               int32_t accumulate, void* workspace, size_t workspace_bytes,
                               const int32_t* rows_z, const int32_t* rows_dy,
                               const int32_t* row_count, int32_t math, int32_t x_is_half,
                               int32_t dy_is_half, void* stream);

/* Adam over every parameter tensor in ONE launch (optax.adam as train_step applies it,
 * snap/trainer.py:236-243: m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2,
 * p -= lr / (1 - b1^step) * m / (sqrt(v / (1 - b2^step)) + eps); `step` counts from 1).
 * `items`: DEVICE table sorted by block_begin; item i owns snap_adam_multi_blocks(n) workgroups
 * of 1024 elements.  p, m, v are updated in place; g is read.  apply_flag (optional DEVICE scalar):
 * the whole update is skipped unless *apply_flag > 0 -- the non-finite step skip of
 * trainer.py:269-276 without a host round trip between the finite check and the update. */
typedef struct SnapAdamItem {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int64_t block_begin;
} SnapAdamItem;
int64_t snap_adam_multi_blocks(int64_t n);
int snap_adam_multi_f32(const SnapAdamItem* items, int32_t n_items, int64_t total_blocks, float lr,
                        float b1, float b2, float eps, int32_t step, const float* apply_flag,
                        void* stream);

/* GroupNorm(+ReLU) backward.  dz: grad w.r.t. the prologue output; add: optional extra
 * gradient summed into dx (identity-residual branch).  mode: SNAP_PRO_GN_RELU /
 * SNAP_PRO_RELU_GN.  dgamma/dbeta [C] (+)=. */
size_t snap_group_norm_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C, int32_t groups);
/* ... _ex: dx_half (optional, [N, HW, C] 2-byte elements) also receives dx rounded to bf16
 * (half_kind = 1) or IEEE half (half_kind = 2), RNE -- the operand image the producing layer's
 * data-gradient convolution (SnapConvExtras.x_half) and kernel-gradient GEMM read instead of the
 * f32 tensor. */
int snap_group_norm_bwd_ex_f32(const float* x, const float* dz, const float* add, float* dx,
                               int32_t N, int32_t HW, int32_t C, int32_t groups,
                               const float* mu, const float* rstd, const float* gamma,
                               const float* beta, int32_t mode, float* dgamma, float* dbeta,
                               int32_t accumulate, void* workspace, size_t workspace_bytes,
                               void* dx_half, int32_t half_kind, void* stream);
/* ... with the statistics pass already done by the convolution that wrote dz (SnapConvExtras.gnb_*):
 * stats = that launch's gn_partial ([N][HW / tile_rows + 2][C][2]), tile_rows = snap_conv2d_tile_rows_ex of
 * its descriptor.  stats == NULL: exactly snap_group_norm_bwd_ex_f32. */
int snap_group_norm_bwd_stats_f32(const float* x, const float* dz, const float* add, float* dx,
                                  int32_t N, int32_t HW, int32_t C, int32_t groups,
                                  const float* mu, const float* rstd, const float* gamma,
                                  const float* beta, int32_t mode, float* dgamma, float* dbeta,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes,
                                  void* dx_half, int32_t half_kind, const float* stats, int32_t tile_rows,
                                  void* stream);
int snap_group_norm_bwd_f32(const float* x, const float* dz, const float* add, float* dx,
                            int32_t N, int32_t HW, int32_t C, int32_t groups,
                            const float* mu, const float* rstd, const float* gamma,
                            const float* beta, int32_t mode, float* dgamma, float* dbeta,
                            int32_t accumulate, void* workspace, size_t workspace_bytes,
                            void* stream);

int snap_weight_standardize_bwd_f32(const float* w, const float* dws, float* dw, int32_t K,
                                    int32_t Cout, float eps, void* stream);
int snap_max_pool_3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int32_t N,
                                int32_t H, int32_t W, int32_t C, void* stream);
/* dprev[N,Hp,Wp,C] = transpose of the bilinear x2 up-sampling applied to dy[N,2Hp,2Wp,C]. */
int snap_upsample2x_bwd_f32(const float* dy, float* dprev, int32_t N, int32_t Hp, int32_t Wp,
                            int32_t C, void* stream);
/* out = dy * [y > 0 iff relu] * row_mask  (epilogue gating; y, row_mask optional). */
int snap_epilogue_bwd_f32(const float* dy, const float* y, const uint8_t* row_mask, float* out,
                          int64_t M, int32_t C, int32_t relu, void* stream);
/* out[C] (+)= column sums of a[M,C]  (bias gradients). */
size_t snap_colsum_workspace_bytes(int64_t M, int32_t C);
/* snap_epilogue_bwd_f32 and the column sums of ITS OUTPUT over the first *row_count rows (all M if
 * NULL) in one pass: the gradient of a bias that sits in front of a ReLU / row mask (layers.py:55-78
 * under jax.grad).  workspace: snap_colsum_workspace_bytes(M, C). */
int snap_epilogue_bwd_colsum_f32(const float* dy, const float* y, const uint8_t* row_mask, float* out,
                                 int64_t M, int32_t C, int32_t relu, const int32_t* row_count,
                                 float* colsum, void* workspace, size_t workspace_bytes, void* stream);
/* ... on 2-byte tensors of the training engine's element type (half_kind 1 = bf16, 2 = IEEE half):
 * dy, y, out [rows, C]; the column sums stay f32.  Rows beyond *row_count are neither read nor
 * written (compact row buffers of the masked MLP). */
int snap_epilogue_bwd_colsum_half(const void* dy, const void* y, void* out, int64_t M, int32_t C,
                                  int32_t relu, const int32_t* row_count, float* colsum,
                                  void* workspace, size_t workspace_bytes, int32_t half_kind,
                                  void* stream);
/* ... plus wsum[C] = sum_r w(r) * out[r, :], w(r) = round_to_element_type(relu?(wsrc[wrows[r] * wstride]))
 * (wrows may be NULL: r itself): the kernel-gradient row of ONE extra input channel of the layer in
 * front -- the 257th channel of the fusion MLP (streetview_encoder.py:277-283) -- taken in the gate
 * pass.  workspace: 2 x snap_colsum_workspace_bytes(M, C). */
int snap_epilogue_bwd_colsum_wsum_half(const void* dy, const void* y, void* out, int64_t M, int32_t C,
                                       int32_t relu, const int32_t* row_count, float* colsum,
                                       void* workspace, size_t workspace_bytes, int32_t half_kind,
                                       const float* wsrc, const int32_t* wrows, int64_t wstride,
                                       int32_t wrelu, float* wsum, void* stream);
/* ... plus (C == 256) the DATA gradient of that channel: dtail[wrows[r] * dstride .. + 3] =
 * (sum_c out[r, c] * round_to_element_type(wtail[c]), 0, 0, 0) -- dtail points at the channel's column of
 * the layer-0 input gradient (f32 rows of dstride floats, 16-byte aligned): with it the data-gradient GEMM
 * in front writes only the whole 128-column tiles (snap_conv2d with Cout_stride > Cout). */
int snap_epilogue_bwd_colsum_wsum_tail_half(const void* dy, const void* y, void* out, int64_t M, int32_t C,
                                            int32_t relu, const int32_t* row_count, float* colsum,
                                            void* workspace, size_t workspace_bytes, int32_t half_kind,
                                            const float* wsrc, const int32_t* wrows, int64_t wstride,
                                            int32_t wrelu, float* wsum, const float* wtail, float* dtail,
                                            int64_t dstride, void* stream);
int snap_colsum_f32(const float* a, int64_t M, int32_t C, float* out, int32_t accumulate,
                    void* workspace, size_t workspace_bytes, void* stream);
/* ... over the listed rows only: sum_{m < *row_count} a[rows[m], :]  (rows / row_count may be NULL). */
int snap_colsum_rows_f32(const float* a, int64_t M, int32_t C, const int32_t* rows,
                         const int32_t* row_count, float* out, int32_t accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);

/* d f_images[B,V,h,w,C] = VJP of snap_lift_pool_f32 w.r.t. f_images (zeroed inside).  The
 * bilinear taps are scatter-added with hardware float atomics: order-dependent sums. */
int snap_lift_pool_bwd_f32(const SnapLiftDesc* desc, const float* f_images, const float* cam,
                           const float* Rt, const float* points, const float* dpooled,
                           float* df_images, void* stream);
/* The same VJP, DETERMINISTIC (bitwise reproducible) and atomic-free: every (voxel, selected
 * view) observation becomes a record, the records are sorted by image pixel (stable radix sort)
 * and one half-wave per pixel gathers the records that touch it in that order; every element of
 * df_images is written exactly once.  workspace: snap_lift_pool_bwd_det_workspace_bytes(desc)
 * bytes (0 = unsupported shape), 256-byte aligned; num_bins <= 32, <= 8 selected views.  Every
 * fusion option of pool_multiview_features (desc->weighted / use_variance / add_minmax; the
 * max / min cotangents are split equally among tied views, as jnp.max's VJP does). */
size_t snap_lift_pool_bwd_det_workspace_bytes(const SnapLiftDesc* desc);
int snap_lift_pool_bwd_det_f32(const SnapLiftDesc* desc, const float* f_images, const float* cam,
                               const float* Rt, const float* points, const float* dpooled,
                               float* df_images, void* workspace, size_t workspace_bytes,
                               void* stream);
/* depth_mlp fusion (streetview_encoder.py:263-267), the VJPs of its two lift passes:
 *   snap_lift_pool_observations_bwd_f32: d pooled -> d obs_feat' [B, N, S, fd] (zero for views that
 *     do not see the point), the pooling VJP on the given (corrected) observations;
 *   snap_lift_observations_bwd_f32: d obs [B, N, S, fd] (the caller's sum of the gradient of
 *     obs[..., :fd] and of obs_feat; depth and ray carry none) -> d f_images, by the deterministic
 *     records / sort / gather machinery with the rows of d obs as the record vectors. */
int snap_lift_pool_observations_bwd_f32(const SnapLiftDesc* desc, const float* cam, const float* Rt,
                                        const float* points, const float* obs_feat,
                                        const float* dpooled, float* dobs, void* stream);
size_t snap_lift_observations_bwd_workspace_bytes(const SnapLiftDesc* desc);
int snap_lift_observations_bwd_f32(const SnapLiftDesc* desc, const float* cam, const float* Rt,
                                   const float* points, const float* dobs, float* df_images,
                                   void* workspace, size_t workspace_bytes, void* stream);
int snap_vertical_pool_bwd_f32(const float* vol, const uint8_t* vvalid, const float* dplane,
                               float* dvol, int64_t M, int32_t Z, int32_t D, int32_t pooling,
                               void* stream);
/* The same VJP for pooling = max from the forward's record (snap_vertical_pool_max_arg_f32): dvol
 * [M, Z, D] is written once and the volume is read only for channels whose maximum is shared by
 * several levels (ties > 1: the gradient is divided between them, as jnp.max's VJP does).  Same
 * values as snap_vertical_pool_bwd_f32, bit for bit. */
int snap_vertical_pool_max_bwd_arg_f32(const float* vol, const uint8_t* vvalid, const uint8_t* argz,
                                       const uint8_t* ties, const float* dplane, float* dvol,
                                       int64_t M, int32_t Z, int32_t D, void* stream);
/* VJP of snap_plane_fuse_match_f32: dplanes[i][M,D] and dy[M,Dm] (gradient w.r.t. the
 * Dense output, for the kernel / bias gradients via wgrad / colsum). */
int snap_plane_fuse_match_bwd_f32(const float* const* planes, const uint8_t* const* valids,
                                  float* const* dplanes, int32_t num_planes, int64_t M,
                                  int32_t D, int32_t pooling, const float* Wm, const float* bm,
                                  int32_t Dm, int32_t normalize, float eps,
                                  const float* dmatching, const float* dfused, float* dy,
                                  void* stream);

/* VJP of snap_pose_score_f32 w.r.t. sim: dsim[B,Nq,X,Y] (planes <= 96 KiB). */
size_t snap_pose_score_bwd_workspace_bytes(int32_t B, int32_t P);
int snap_pose_score_bwd_f32(const float* dscores, const float* poses, const float* q_xy,
                            const uint8_t* valid_q, const uint8_t* map_valid, int32_t B,
                            int32_t Nq, int32_t X, int32_t Y, int32_t P, float cell_size,
                            int32_t mask_oob, float* dsim, void* workspace,
                            size_t workspace_bytes, void* stream);
/* ... with flags: bit 0 = accumulate the score planes with float LDS atomics (order-dependent sums;
 * A/B timing).  Default (0): where the 8-byte plane fits the LDS (X * Y <= 19 456 cells, no
 * out-of-bounds mask, P <= 10 240) the plane is accumulated in 64-bit fixed point -- exactly
 * associative, so d sim is bitwise reproducible. */
int snap_pose_score_bwd_ex_f32(const float* dscores, const float* poses, const float* q_xy,
                               const uint8_t* valid_q, const uint8_t* map_valid, int32_t B,
                               int32_t Nq, int32_t X, int32_t Y, int32_t P, float cell_size,
                               int32_t mask_oob, int32_t flags, float* dsim, void* workspace,
                               size_t workspace_bytes, void* stream);
/* In place: dsim <- dsim * [sim > 0 iff clip] * coef[b]; partial[B,num_partial] <- partial
 * sums of dsim*sim (temperature gradient). */
int snap_sim_bwd_prepare_f32(float* dsim, const float* sim, int32_t B, int64_t per_scene,
                             int32_t clip_negative, const float* coef, float* partial,
                             int32_t num_partial, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* SNAP_HIP_H_ */
