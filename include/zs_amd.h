/* zs_amd.h -- C ABI of libzs_amd.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * ASR-TTS autoencoder hot path of andi611/ZeroSpeech-TTS-without-T.
 *
 * The reference has no FFI / plugin seam (it is pure Python on torch.nn); this ABI is the inner
 * seam of the drop-in Python surface (zs_amd.model / zs_amd.trainer / zs_amd.convert).  Each entry
 * point states which reference arithmetic it replaces (file:line into the reference repository).
 *
 * Conventions
 *  - Every function returns 0 on success or a negative ZS_E* code; zs_last_error() gives text.
 *    Nothing is thrown across the boundary.
 *  - The caller owns every buffer (including workspaces).  The library never allocates, frees or
 *    synchronises; every call only enqueues kernels on `stream` (hipStream_t passed as void*), so
 *    calls are stream-ordered, re-entrant and hipGraph-capturable.
 *  - Activations are channels-last: a tensor [B, T, C] is stored as B*T rows of `ld` elements
 *    (ld >= C, ld a multiple of 32 elements unless stated).  Columns [C, ld) of an activation that
 *    feeds a GEMM must hold zeros; every kernel here writes them as zeros.
 *  - dtype: ZS_F32 (exact-fp32 path on v_mfma_f32_32x32x2_f32) or ZS_BF16 (bf16 storage,
 *    v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Statistics, gradients of parameters, optimizer
 *    state and losses are always fp32.
 */
#ifndef ZS_AMD_H
#define ZS_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZS_ABI_VERSION 1

enum { ZS_F32 = 0, ZS_BF16 = 1 };
enum { ZS_OK = 0, ZS_EINVAL = -1, ZS_ELAUNCH = -2, ZS_EWORKSPACE = -3 };
enum { ZS_PAD_ZERO = 0, ZS_PAD_REFLECT = 1 };
enum { ZS_ACT_NONE = 0, ZS_ACT_LRELU = 1, ZS_ACT_SIGMOID = 2, ZS_ACT_TANH = 3 };
enum { ZS_STORE_ROWS = 0, ZS_STORE_SPLIT2 = 1 };   /* SPLIT2 = pixel_shuffle_1d by 2 (see below) */
enum { ZS_RES_NONE = 0, ZS_RES_IDENTITY = 1, ZS_RES_AVGPOOL2 = 2, ZS_RES_UPSAMPLE2 = 3 };

int zs_abi_version(void);
const char* zs_last_error(void);
/* Kernel-selection knobs (defaults also read from the environment at first use):
 *   "gemm_dma" (ZS_GEMM_DMA, 1): LDS-DMA operand staging; 0 = register-staged kernel
 *   "gemm_ring" (ZS_GEMM_RING, 1): allow the 256x128 3-stage-ring kernel
 *   "gemm_ring_min_tiles" (ZS_GEMM_RING_MIN_TILES, 256): use it when the problem has at least that many 256x128 tiles
 *   "gemm_pp" (ZS_GEMM_PP, 1): ping-pong schedule of the ring kernel (two wave groups one barrier apart)
 *   "gemm_p8" (ZS_GEMM_P8, 1) / "gemm_p8_min_tiles" (ZS_GEMM_P8_MIN_TILES, 200): 256x256 quadrant ping-pong kernel when the
 *       packed weight has a multiple of 256 rows and the problem has at least that many 256x256 tiles
 *   "wgrad_p8" (ZS_WGRAD_P8, 1): 256x256 ping-pong weight-gradient kernel (bf16) where its heuristics accept the shape;
 *       0 = always the 128x128 kernel, 2 = whenever bf16 (tests)
 *   "gru_persist" (ZS_GRU_PERSIST, 1): one persistent launch per GRU pass (forward and BPTT: the time loop runs on the
 *       device, W_hh stays in registers, h / dgh travel between workgroups as data-tagged 8-byte granules) when the grid fits
 *       one workgroup per CU and H allows it; 0 = one launch per time step
 *   "gru_wide" (ZS_GRU_WIDE, 1): bf16, H in {128, 256, 512}: the persistent kernels in their 16-row x 64-unit tiling (a wave
 *       owns 16 units over the full K, W_hh slice in registers, one barrier per step, h / dgh of the group in an LDS image);
 *       0 = the 32 x 32 (forward) / 16 x 64 K-split (BPTT) tilings, which also serve fp32 and other H
 *   "gru_spin_limit" (ZS_GRU_SPIN_LIMIT, 2^21): granule sweeps a persistent GRU wave makes before it gives up on its group
 *       (sets the status word, see ZsGruFwd.status); 0 forces the timeout path (tests)
 * The knobs are process-wide atomics: they may be set from any thread; a launch reads each knob once.
 * Returns the previous value, or ZS_EINVAL for an unknown key. */
int zs_set_option(const char* key, int value);

/* ---------------------------------------------------------------------------------------------
 * zs_gemm_conv: implicit-GEMM Conv1d / Linear / transposed-conv (data gradient) on MFMA.
 *   out[m, n] = epilogue( sum_{j<taps} sum_{ci<cin_pad} A[row(m, j), ci] * W[n, j*cin_pad + ci] )
 *   m = b*T_out + t.  gather 0 (forward):  row = reflect_or_zero(t*stride + j - pad_left) in [0,T_in)
 *                     gather 1 (dgrad):    u = t; s = u - j; row = s/stride if s>=0, s%stride==0,
 *                                          s/stride < T_in, else a zero row.
 * Replaces: pad_layer()+nn.Conv1d (model/model.py:20-40), linear() (model/model.py:69-78), the GRU
 * input/recurrent products of nn.GRU (model/model.py:59-66), and their autograd data gradients.
 * Epilogue order: +bias[n] -> +pre_vec[vec_idx[b]][n] -> act (-> colsum) -> *lrelu'(dact_src[m][n]) ->
 *                 +add_src[m][n] -> store out (-> out2 = value + vec2[vec_idx[b]][col]).
 * ZS_STORE_SPLIT2 realises pixel_shuffle_1d (model/model.py:43-51) for weights whose output
 * channels were packed as n' = r*(N/2) + c:  (m, n') -> row 2m + (n' >= N/2), col n' mod (N/2).
 * `groups` > 1 launches independent problems that differ by the *_gstride element offsets.
 */
typedef struct {
  int32_t dtype;
  const void* A; int64_t lda; int64_t a_batch_stride;
  int32_t B, T_in, T_out;
  int32_t taps, stride, pad_left, pad_mode, gather;
  int32_t cin_pad;                 /* K per tap: a whole number of 128-byte chunks (multiple of 64 bf16 / 32 fp32).
                                      A rows are read up to cin_pad columns: bytes past C must be finite (zeros or the
                                      next row), the matching W columns are zero */
  const void* W; int64_t ldw;      /* packed [n_pad][ldw], ldw >= taps*cin_pad, zero padded */
  int32_t N, n_pad;                /* valid columns; packed rows (multiple of 128) */
  const float* bias;
  const float* pre_vec; int64_t pre_vec_ld; const int64_t* vec_idx;
  int32_t act; float slope;
  const void* dact_src; int64_t dact_ld;
  const void* add_src; int64_t add_ld; int32_t add_f32;
  void* out; int64_t ldc; int32_t out_f32; int32_t out_cols; int32_t store_mode;
  void* out2; int64_t ldc2; int32_t out2_cols; int32_t store_mode2;
  const float* vec2; int64_t vec2_ld;
  int32_t groups; int64_t a_gstride, w_gstride, out_gstride, bias_gstride;
  /* optional per-sample column sums of the value BEFORE dact_src / add_src (the per-sample part of nn.Embedding's backward
   * where an embedding row was added to this GEMM's input, model/model.py:319,336,353; finished by zs_emb_scatter):
   *   colsum[b][n - colsum_col0] += sum_t value[b*T_out + t][n]      for colsum_col0 <= n < N   (colsum_post: of `out`)
   * One owner per (b, n): plain read-modify-write, no atomics.  Needs T_out | 128 (whole samples per tile). */
  float* colsum; int64_t colsum_ld; int32_t colsum_col0;
  int32_t colsum_post;             /* 1: sum the STORED value (after dact_src / add_src) instead */
  /* ragged batch (inference: the tail fragments of convert.py:154-165 have 128..254 frames each and the reference runs them one
   * by one): sample b has lengths[b] <= T_in valid input rows and (lengths[b] + pad_left + pad_right - taps) / stride + 1 valid
   * output rows; the padding (reflect / zero) is applied at EACH sample's own end, exactly as if the sample ran alone.  Output
   * rows past a sample's length are computed from zero rows (finite, never read by a length-aware consumer).  T_in / T_out stay
   * the row strides of the batch.  gather 0 only; B < 65536, T_in < 32768.  null = every sample has T_in rows. */
  const int32_t* lengths; int32_t pad_right;
  /* 2-D convolution (w_in > 0; nn.Conv2d of the stage-2 critic, model/model.py:113-173, on channels-last [B, H*W, C] rows): a
   * sample's T_in rows are an (T_in / w_in) x w_in image, its T_out output rows an (T_out / w_out) x w_out image, tap
   * j = kw*taps_h + kh (kw = j / taps_h along the row, kh = j % taps_h across rows: the order of nn.Conv2d's weight [co][ci][kw][kh]
   * when W is the frequency axis); the gather rule above applies per axis with the same stride, pad_left and pad_mode.
   * 0 = the 1-D convolution over T.  Not with lengths. */
  int32_t w_in, w_out, taps_h;
} ZsGemmConv;
int zs_gemm_conv(const ZsGemmConv* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_gemm_wgrad: weight gradient of the same convolution,
 *   dW[co, ci, j] = sum_m dY[m, co] * X[row(m, j), ci]        (row() as gather 0 above)
 * computed as split-K partial slabs [split][co][j][ci_pad] in `workspace` (plain coalesced stores),
 * then reduced in a fixed order (bitwise reproducible) into dW with element strides (so, si, sj):
 * Conv1d weight [Cout,Cin,k]: (Cin*k, k, 1); Linear weight [out,in]: (in, 1, 0).
 * co_split2 un-permutes the SPLIT2 output-channel packing.  accumulate != 0 adds to dW / db.
 * No atomics anywhere: results are bitwise reproducible run to run.
 * Replaces: autograd's convolution_backward / addmm weight gradients under loss.backward()
 * (trainer.py:330).
 */
typedef struct {
  int32_t dtype;
  const void* dY; int64_t ldy; int32_t y_cols;     /* y_cols: readable columns of dY rows (<= ldy) */
  const void* X; int64_t ldx; int64_t x_batch_stride; int32_t x_cols;
  int32_t B, T_in, T_out;
  int32_t taps, stride, pad_left, pad_mode;
  int32_t Cout, Cin;
  float* dW; int64_t so, si, sj;
  float* db;                       /* optional bias gradient [Cout]: db[co] (+)= sum_m dY[m, co], same slab/reduce path */
  int32_t co_split2; int32_t accumulate;
  int32_t splits;                  /* 0 = choose */
  void* workspace; size_t workspace_bytes;
  int32_t w_in, w_out, taps_h;     /* 2-D convolution: as in ZsGemmConv (dW[co, ci, j], j = kw*taps_h + kh); 0 = 1-D */
} ZsGemmWgrad;
size_t zs_gemm_wgrad_workspace_bytes(const ZsGemmWgrad* p);
int zs_gemm_wgrad(const ZsGemmWgrad* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_pack_weight: fp32 parameter (PyTorch layout, strides so/si/sj as above) -> GEMM operand.
 * Writes the n_rows x n_cols block of dst that starts at (row_offset, col_offset), dst pitch ldw:
 *   transpose 0 (forward):  block[n][j*inner_pad + ci] = W[co(n)][ci][j]
 *   transpose 1 (dgrad):    block[ci][j*inner_pad + n] = W[co(n)][ci][j]
 * co(n) = n, or with co_split2: n < Cout/2 ? 2n : 2(n - Cout/2) + 1.  Everything else in the block is
 * zero (rows >= Cout|Cin, inner index >= Cin|Cout, taps >= taps).
 */
typedef struct {
  int32_t dtype;
  const float* W; int64_t so, si, sj;
  int32_t Cout, Cin, taps;
  int32_t transpose, co_split2;
  int32_t inner_pad;               /* cin_pad (transpose 0) or cout_pad (transpose 1) */
  void* dst; int64_t ldw;
  int32_t n_rows, n_cols, row_offset, col_offset;
  const int32_t* row_perm;         /* optional: output row n reads parameter row row_perm[n] (overrides co_split2) */
} ZsPackWeight;
int zs_pack_weight(const ZsPackWeight* p, void* stream);
/* the same for n jobs (one dtype) in ceil(n/32) launches: the per-step repack of a whole net after the optimiser step.
 * max_blocks > 0 caps the grid (the workgroups then walk the tile list): a re-pack that runs on a side stream beside a chain of
 * small kernels must leave them wave slots (about two workgroups per CU still stream at the HBM rate); 0 = one workgroup per tile */
int zs_pack_weight_batch(const ZsPackWeight* jobs, int32_t n, int32_t max_blocks, void* stream);

/* zs_copy_vec_batch: n small fp32 vector copies in ONE launch, dst[i*dst_stride] = src[i*src_stride] for i < len: the bias
 * re-ordering that goes with the pixel-shuffle packing (co_split2) and the stacked nn.GRU biases, issued with the re-pack. */
typedef struct {
  const float* src; float* dst;
  int32_t len, src_stride, dst_stride;
} ZsVecCopy;
int zs_copy_vec_batch(const ZsVecCopy* jobs, int32_t n, void* stream);

/* zs_cast_rows: dst[r][col_off + c] = f(src[r][c]) for c < cols, zeros for cols <= c < fill_cols.
 * f = identity or leaky_relu (model/model.py:446).  src fp32 or T (src_f32), dst T or fp32. */
typedef struct {
  int32_t dtype;
  const void* src; int64_t ld_src; int32_t src_f32;
  void* dst; int64_t ld_dst; int32_t dst_f32; int32_t col_off;
  int64_t rows; int32_t cols, fill_cols;
  int32_t act; float slope;
  /* optional second destination from the same read (type T): dst2[r][col_off2 + c] = f2(src[r][c]), zeros up to fill_cols2
   * (the encoder's input goes to the conv bank raw and to the concatenation through leaky_relu, model/model.py:441-446) */
  void* dst2; int64_t ld_dst2; int32_t col_off2, fill_cols2; int32_t act2; float slope2;
} ZsCastRows;
int zs_cast_rows(const ZsCastRows* p, void* stream);

/* zs_add_rowvec: out[b,t,c] = x[b,t,c] + vec[idx[b]][c]  (x + emb.view(B,C,1), model/model.py:319,336,353)
 * x may be null: broadcast (append_emb, model/model.py:81-85). */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  const float* vec; int64_t vec_ld; const int64_t* idx;
  void* out; int64_t ldo; int32_t B, T, C, fill_cols;   /* C: valid columns of vec; fill_cols: columns written */
} ZsAddRowvec;
int zs_add_rowvec(const ZsAddRowvec* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_instnorm_fwd: nn.InstanceNorm1d (no affine, biased var, eps) over T for every (b, c), fused with
 * nn.Dropout and the block residual (model/model.py:416-438, 317-342):
 *   out = dropout(xhat) + residual ;  out2 = out + vec2[idx[b]]   (optional)
 * Dropout keep-mask: `mask` (uint8 [B,T,mask_ld]) if given, else counter-hash RNG(seed, stream_id)
 * that zs_instnorm_bwd regenerates.  T <= 256.
 */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  void* out; int64_t ldo;
  void* out2; int64_t ldo2; const float* vec2; int64_t vec2_ld; int32_t vec2_cols; const int64_t* idx;
  float* mean; float* rstd;       /* [B][C] fp32, saved for backward (may be null) */
  int32_t B, T, C; float eps;
  float drop_p; uint64_t seed; uint32_t stream_id; const uint8_t* mask; int64_t mask_ld;
  int32_t res_mode; const void* res; int64_t ldres; int32_t T_res; int32_t res_pad_mode;
  const uint64_t* seed_ptr;      /* optional device scalar added to `seed` (graph replay: see zs_step_counters) */
  /* ragged batch: the statistics of sample b run over its first lengths[b] <= T rows (T stays the row stride); rows past the
   * length are written as zeros.  res_lengths[b] <= T_res: valid rows of the ZS_RES_AVGPOOL2 residual (its odd-length
   * reflect / zero pad, model/model.py:424, happens at the sample's own end).  null = T / T_res for every sample. */
  const int32_t* lengths; const int32_t* res_lengths;
  /* 1: mean / rstd are INPUTS (statistics computed elsewhere, e.g. by the producing kernel): the two reductions over T are skipped */
  int32_t stats_given;
} ZsInstNormFwd;
int zs_instnorm_fwd(const ZsInstNormFwd* p, void* stream);

/* zs_instnorm_bwd: gradient w.r.t. the PRE-activation of the leaky_relu that feeds the norm:
 *   g = dout * keep/(1-p);  dx = rstd*(g - mean_t(g) - xhat*mean_t(g*xhat));  dz = dx * lrelu'(x). */
typedef struct {
  int32_t dtype;
  const void* dout; int64_t ldd;
  const void* x; int64_t ldx;
  const float* mean; const float* rstd;
  void* dz; int64_t ldz;
  int32_t B, T, C;
  float drop_p; uint64_t seed; uint32_t stream_id; const uint8_t* mask; int64_t mask_ld;
  float slope;
  const uint64_t* seed_ptr;
} ZsInstNormBwd;
int zs_instnorm_bwd(const ZsInstNormBwd* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_grad_combine: everything autograd does between two GEMMs on the backward path:
 *   v[b,t,c]  = fold(gp)[b,t,c]                 gp = zs_gemm_conv(gather=1) output over the padded
 *                                                length T+pad_left+pad_right; reflect pads fold back
 *   emb_sum[b][c] += sum_t v[b,t,c]              (optional; per-sample part of nn.Embedding backward,
 *                                                 finished by zs_emb_scatter -- no atomics)
 *   v += residual term of `res` (IDENTITY: res[t]; AVGPOOL2: 0.5*res[t/2]; UPSAMPLE2: res[2t]+res[2t+1])
 *   v *= lrelu'(dact_src)                         (optional)
 *   store: rows, or un-pixel-shuffle (unshuffle=1): (b,t,c) -> row (b, t/2), col (t&1)*C + c
 */
typedef struct {
  int32_t dtype;
  const void* gp; int64_t ldg; int32_t pad_left, pad_right, pad_mode;
  int32_t B, T, C;
  float* emb_sum; int64_t emb_ld; int32_t emb_cols;
  int32_t res_mode; const void* res; int64_t ldres;
  const void* dact_src; int64_t dact_ld; float slope;
  void* out; int64_t ldo; int32_t unshuffle;
} ZsGradCombine;
int zs_grad_combine(const ZsGradCombine* p, void* stream);

/* zs_emb_scatter: nn.Embedding weight gradient from per-sample sums, in fixed sample order:
 *   demb[r][c] (+)= sum_{b : idx[b] == r} emb_sum[b][c]      (model/model.py:310-314 lookups) */
typedef struct {
  const float* emb_sum; int64_t emb_ld; const int64_t* idx; int32_t B;
  float* demb; int64_t demb_ld; int32_t n_rows, C; int32_t accumulate;
} ZsEmbScatter;
int zs_emb_scatter(const ZsEmbScatter* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_mbv_fwd / zs_mbv_bwd: Multilabel-Binary-Vector discretiser (model/model.py:93-110, 474-480).
 * logits rows [B*T'][ld] hold E (l0,l1) pairs.  noise_kind 0: Gumbel noise G given (bit-exact
 * contract); 1: uniform U given, g = -log(-log(U+1e-20)+1e-20) on device; 2: U from the
 * counter-hash RNG(seed).  bits = ((hard - y0) + y0) with y = softmax((l+g)/tau), ties -> 1.0
 * (index 0).  y0 is saved for backward: dl0 = dbits*y0*(1-y0)/tau, dl1 = -dl0.
 */
typedef struct {
  int32_t dtype;
  const void* logits; int64_t ld; int32_t logits_f32;
  const float* noise; int32_t noise_kind; uint64_t seed;
  int64_t rows; int32_t E; float tau;
  void* bits; int64_t ld_bits; int32_t bits_fill_cols;
  float* bits_f32;               /* optional compact [rows][E] fp32 copy (API output) */
  float* y0;                     /* optional [rows][E] */
  const uint64_t* seed_ptr;      /* optional device scalar added to `seed` */
} ZsMbvFwd;
int zs_mbv_fwd(const ZsMbvFwd* p, void* stream);
typedef struct {
  int32_t dtype;
  const void* dbits; int64_t ld_dbits;
  const float* y0; int64_t rows; int32_t E; float tau;
  void* dlogits; int64_t ld; int32_t fill_cols;
} ZsMbvBwd;
int zs_mbv_bwd(const ZsMbvBwd* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_gru_fwd / zs_gru_bwd: bidirectional single-layer nn.GRU with zero initial state
 * (model/model.py:59-66, 389, 299), gate order r,z,n.  gi = X W_ih^T + b_ih for BOTH directions
 * ([B*T][ldgi], columns dir*3H + {r,z,n}*H) comes from zs_gemm_conv.  Per time step the library
 * enqueues one grouped recurrent product (zs_gemm_conv kernel, 2 directions) and one gate kernel.
 * out[b,t, out_col + dir*H + j] = h_t.  gates saves r,z,n,(W_hn h + b_hn) for backward.
 * Persistent path (default when (H/32)*ceil(B/32)*2 <= #CUs and H is a multiple of 128 (bf16) / 64 (fp32), H <= 512 /
 * 256): ONE launch per pass, see gru_persist_fwd_kernel; needs the interleaved packing below.
 * Fast path (H % 32 == 0 and whh_interleaved): ONE launch per step -- a 16-rows-per-wave MFMA product whose
 * fragments come straight from L2 (no LDS, no barrier) with the gate math fused in its epilogue.  It needs the
 * rows of whh packed gate-interleaved per 32 hidden units: row hc*96 + g*32 + u = W_hh[g*H + hc*32 + u].
 */
typedef struct {
  int32_t dtype;
  int32_t B, T, H;
  const void* gi; int64_t ldgi;
  const void* whh; int64_t ldw; int32_t n_pad; int64_t w_gstride;   /* packed fwd [2][n_pad][ldw], N=3H, K=H */
  const float* bhh; int64_t bhh_gstride;                           /* [2][3H] */
  void* out; int64_t ldo; int32_t out_col;
  void* gates; /* T dtype [B][T][2][4H] or null (inference) */
  float* work; size_t work_bytes;   /* >= zs_gru_work_bytes */
  int32_t whh_interleaved;
  uint32_t* status;              /* optional caller-owned device word, STICKY: a persistent pass whose bounded spin expires ORs
                                    bit 0 (forward) / bit 1 (BPTT) into it and goes on with invalid data; the library never
                                    clears it.  The caller reads it at its own host sync points (Trainer.train: every logged
                                    loss; convert/encode: the end of a batch) and raises. */
  /* optional broadcast riding on the recurrence's stores (append_emb, model/model.py:81-85,357: the third block of the
   * concatenation the GRU output goes into): out[b][t][bcast_col + c] = bcast_vec[bcast_idx[b]][c] for c < 2H and every t.
   * The persistent kernel is latency-bound and leaves HBM idle, so these 64 MB cost nothing there. */
  const float* bcast_vec; int64_t bcast_ld; const int64_t* bcast_idx; int32_t bcast_col;
  /* 1: direction 1 also steps through t = 0 .. T-1 (instead of T-1 .. 0).  Ragged batches: the caller reverses the rows of
   * every sample over ITS OWN length (zs_rows_reverse) in the direction-1 columns of gi before and of out after this call, so
   * both directions are forward recurrences whose first lengths[b] steps are exactly the sample's own sequence. */
  int32_t dir1_forward;
} ZsGruFwd;
size_t zs_gru_work_bytes(int32_t B, int32_t H);
int zs_gru_fwd(const ZsGruFwd* p, void* stream);
/* test / debug hook (synchronises the stream): ZS_OK unless the last persistent pass over `work` had a workgroup give up
 * waiting for its group -- bounded spins instead of a hang; the results of that pass are then invalid */
int zs_gru_check(const float* work, int32_t B, int32_t H, void* stream);

/* zs_rows_reverse: in-place reversal of the rows of every sample over its own length, restricted to a column block:
 *   x[b][t][col0 + c] <-> x[b][lengths[b] - 1 - t][col0 + c]   for t < lengths[b] / 2, c < cols   (rows b*T + t, pitch ld elements)
 * (nn.GRU's reverse direction on a ragged batch, see ZsGruFwd.dir1_forward).  cols * element size must be a multiple of 4. */
typedef struct {
  int32_t dtype;
  void* x; int64_t ld; int32_t col0, cols;
  int32_t B, T; const int32_t* lengths;
} ZsRowsReverse;
int zs_rows_reverse(const ZsRowsReverse* p, void* stream);
typedef struct {
  int32_t dtype;
  int32_t B, T, H;
  const void* dout; int64_t ldd; int32_t dout_col;   /* gradient w.r.t. out columns */
  const void* out; int64_t ldo; int32_t out_col;     /* forward outputs (h_t) */
  const void* gates;
  const void* whh_t; int64_t ldw; int32_t n_pad; int64_t w_gstride; /* packed transposed [2][n_pad][ldw], N=H, K=3H */
  void* dgi; int64_t ldgi;       /* [B*T][ldgi] cols dir*3H.. : grad w.r.t. gi (input projections) */
  void* dgh; int64_t ldgh;       /* same shape: grad w.r.t. (h W_hh^T + b_hh) */
  float* work; size_t work_bytes;
  uint32_t* status;              /* sticky status word, see ZsGruFwd */
} ZsGruBwd;
int zs_gru_bwd(const ZsGruBwd* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * zs_l1_loss: loss = mean|x_dec - x| over rows*F (trainer.py:328) and the gradient w.r.t. the
 * pre-sigmoid logits: g = sign(x_dec - x)/(rows*F) * x_dec*(1-x_dec).
 * loss_out receives the scalar (two-pass, fixed order).  partial: >= 1024 floats.
 */
typedef struct {
  int32_t dtype;
  const float* x_dec; int64_t ld_dec;
  const float* x; int64_t ldx;
  int64_t rows; int32_t F;
  void* dlogits; int64_t ldg; int32_t fill_cols;   /* may be null (loss only) */
  float* partial; float* loss_out; float grad_scale;
} ZsL1Loss;
int zs_l1_loss(const ZsL1Loss* p, void* stream);

/* zs_sqnorm: out[0] = sum g[i]^2 (double accumulation, fixed order).  partial: >= 1024 doubles. */
int zs_sqnorm(const float* g, int64_t n, double* partial, float* out_sq, void* stream);

/* zs_adam_clip: nn.utils.clip_grad_norm_(max_norm) (utils.py:53-55) fused with torch.optim.Adam's
 * single-tensor update (trainer.py:66,332): coef = min(1, max_norm/(sqrt(sumsq)+1e-6)); g *= coef;
 * m = m + (g-m)(1-b1); v = v*b2 + g*g*(1-b2); p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
 * sumsq is a device scalar (from zs_sqnorm); max_norm <= 0 disables clipping.
 * grad_scale (0 = 1): g holds grad_scale^-1 times the gradient -- the SUM over `world` data-parallel ranks with
 * grad_scale = 1/world -- so the average is never materialised: norm = grad_scale*sqrt(sumsq), g_eff = g*grad_scale*coef. */
typedef struct {
  float* p; float* g; float* m; float* v; int64_t n;
  float lr, beta1, beta2, eps, bc1, bc2;
  const float* sumsq; float max_norm; int32_t write_clipped_grad;
  const int32_t* step_ptr;       /* optional device step count t: bc1 = 1-beta1^t, bc2 = 1-beta2^t computed on device */
  float grad_scale;
  int32_t max_blocks;            /* > 0: grid cap (see zs_pack_weight_batch); 0 = default */
} ZsAdam;
int zs_adam_clip(const ZsAdam* p, void* stream);

/* zs_host_fetch: dst[0..bytes) = src[0..bytes) by a KERNEL that reads pinned (device-accessible) host memory over PCIe with
 * 16-byte loads, on a deliberately small grid (`workgroups`, 0 = 32; 8 loads in flight per lane cover the link's
 * bandwidth-latency product).  It replaces the H2D memcpy of the next batch (trainer.py:238-244) inside the captured step: on
 * this stack a memcpy node of a hipGraph is not overlapped with the graph's kernel nodes, a kernel node on a forked stream is.
 * bytes must be a multiple of 16, both pointers 16-byte aligned. */
int zs_host_fetch(const void* src, void* dst, size_t bytes, int32_t workgroups, void* stream);

/* zs_step_counters: per-step device-side state for hipGraph replay (kernel arguments are frozen in a graph):
 * *seed += 0x9E3779B97F4A7C15, *step += 1.  Either pointer may be null. */
int zs_step_counters(uint64_t* seed, int32_t* step, void* stream);

/* zs_softmax_ce: nn.CrossEntropyLoss (mean) + gradient (trainer.py:297-304). logits fp32 [B][ld]. */
typedef struct {
  const float* logits; int64_t ld; const int64_t* target; int32_t B, n_class;
  float* loss_out; float* dlogits; int64_t ldg; float grad_scale; int32_t* correct_out;
} ZsSoftmaxCE;
int zs_softmax_ce(const ZsSoftmaxCE* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Vocoder (convert.py:39-62): batched Griffin-Lim on 1024-point real FFTs held in LDS.
 * mag: fp32 [n_utt][T_max][513] linear magnitudes (zero rows past each utterance's length).
 * zs_gl_istft: frames of spec (complex64 [n_utt][T_max][513]) -> windowed irfft -> overlap-add /
 *              window-sum-square -> wav [n_utt][wav_ld] (200*(T-1) samples, centre-trimmed).
 * zs_gl_stft_project: wav -> reflect-padded frames -> rfft -> spec = mag * E/max(1e-8,|E|).
 */
typedef struct {
  const float* spec; const float* mag; const int32_t* lengths; int32_t n_utt, T_max;
  float* wav; int64_t wav_ld; float* frames_ws;    /* frames_ws: [n_utt][T_max][1024] fp32 scratch */
} ZsGlIstft;
int zs_gl_istft(const ZsGlIstft* p, void* stream);
typedef struct {
  const float* wav; int64_t wav_ld; const float* mag; const int32_t* lengths; int32_t n_utt, T_max;
  float* spec;
} ZsGlStft;
int zs_gl_stft_project(const ZsGlStft* p, void* stream);
/* zs_gl_iter: ONE Griffin-Lim iteration (convert.py:46-50) as one fused kernel per utterance tile:
 *     x = istft(spec_in);  E = stft(x);  spec_out = mag * E / max(1e-8, |E|)
 * (the windowed frames and the waveform never leave LDS; 512-point complex FFT per wave for the real 1024-point transforms).
 * spec_out == NULL: the final pass -- only x = istft(spec_in), written to wav [n_utt][wav_ld] (200*(T-1) samples each).
 * spec_in / spec_out: complex64 [n_utt][T_max][513], distinct buffers (neighbouring tiles read spec_in while spec_out is
 * written).  Utterances need lengths[u] >= 4 frames (the reflect padding of 512 samples must fit; shorter ones are skipped:
 * the reference's decoder never emits fewer than 16).
 * zs_griffin_lim: the whole loop of convert.py:39-52 from one call: spec_a holds X0 = S (zero phase) on entry; n_iter
 * iterations ping-pong spec_a / spec_b; then the final inverse pass into wav. */
typedef struct {
  const float* spec_in; float* spec_out;
  const float* mag; const int32_t* lengths; int32_t n_utt, T_max;
  float* wav; int64_t wav_ld;
  int32_t tile_frames;           /* STFT frames per workgroup tile (4..42), 0 = default 10 */
  const int32_t* host_lengths;   /* optional copy of `lengths` in HOST memory: zs_griffin_lim then cuts its launch chains at equal frame
                                    counts instead of equal utterance counts (utterances stay in order) */
} ZsGlIter;
int zs_gl_iter(const ZsGlIter* p, void* stream);
int zs_griffin_lim(const ZsGlIter* p, float* spec_a, float* spec_b, int32_t n_iter, void* stream);
/* spec_a / spec_b are SCRATCH for zs_griffin_lim ([n_utt][T_max][513] complex64 each, contents on entry ignored): the loop starts
 * from X0 = S with zero phase (convert.py:41), which the first iteration reads from `mag` itself. */
/* zs_griffin_lim issues the loop as up to `gl_chains` (zs_set_option, default 3) independent launch chains over contiguous utterance
 * ranges, on internal streams measured (once per process) to execute side by side, joined back into `stream` before it returns:
 * utterances do not depend on each other, and a chain's partly filled last round of workgroups is filled by the others.  One chain
 * for small batches (< 8192 frames per chain) and under stream capture.  zs_gl_chains_used: chains of the last call (reporting). */
int zs_gl_chains_used(void);
/* spectrogram2wav pre/post (convert.py:56-60): de-normalise to amplitude; de-preemphasis IIR. */
/* ---------------------------------------------------------------------------------------------
 * Feature extraction feeding the path (preprocess.py:227-258 get_spectrograms after the host-side
 * librosa.effects.trim): wav [n_utt][wav_ld] fp32, n_samples[u] valid samples ->
 *   mag[u][t][0..513) = clip((20 log10(max(1e-5, |STFT(preemph(y))|)) - ref_db + max_db) / max_db, 1e-8, 1), t < 1 + n_samples/200
 *   amp (optional, [n_utt][T_max][513]): the linear magnitudes, input of zs_pre_mel.
 * STFT as in the vocoder: n_fft 1024, hop 200, hann(800) centred, reflect-padded centre frames.
 */
typedef struct {
  const float* wav; int64_t wav_ld; const int32_t* n_samples; int32_t n_utt, T_max;
  float preemph, ref_db, max_db;
  float* mag; int64_t mag_ld;      /* [n_utt][T_max][mag_ld], mag_ld >= 513 */
  float* amp;
} ZsPreSpec;
int zs_pre_spectrogram(const ZsPreSpec* p, void* stream);
/* mel[r][m] = clip((20 log10(max(1e-5, sum_k basis[m][k] amp[r][k])) - ref_db + max_db) / max_db, 1e-8, 1); basis [n_mels][513]
 * (librosa.filters.mel(sr, n_fft, n_mels), preprocess.py:243-252) */
int zs_pre_mel(const float* amp, const float* basis, float* mel, int64_t rows, int32_t n_mels, float ref_db, float max_db, void* stream);
int zs_gl_denormalize(const float* mag_norm, float* mag_amp, int64_t n, void* stream);
int zs_gl_deemphasis(float* wav, int64_t wav_ld, const int32_t* lengths, int32_t n_utt, float coef, void* stream);
/* zs_gl_frame_mse: the frame statistics of librosa.effects.trim (convert.py:61; librosa's rms on the centred, reflect-padded
 * signal): mse[u][f] = mean_{i < frame_length} y_u[reflect(f*hop + i - frame_length/2)]^2, f <= L_u / hop, L_u = 200 *
 * (lengths[u] - 1) samples of wav row u (needs L_u > frame_length / 2), accumulated in double.  The caller thresholds at
 * top_db below the maximum and slices: only these few hundred values per utterance are examined on the host. */
int zs_gl_frame_mse(const float* wav, int64_t wav_ld, const int32_t* lengths, int32_t n_utt, int32_t frame_length, int32_t hop,
                    double* mse, int64_t mse_ld, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Stage 2 (model/model.py:113-228, trainer.py:257-294, 467-560, utils.py:58-77): PatchDiscriminator /
 * TargetClassifier and the WGAN-GP penalty.
 * Activations of the 2-D nets are channels-last [B, H, W, C] = B*H*W rows of ld elements with H = time and W = frequency, so
 * the loader's [B, T, F] spectrogram IS the input [B, H=T, W=F, C=1] (the reference's Conv2d sees [B, 1, F, T]: its kernel
 * index along F is the tap axis here, the one along T the gather axis).  A k x k / stride-s Conv2d (pad_layer is_2d) runs as
 *   zs_conv2d_gather  : im2col along H only, Xh[(b,ho), w, kh*C + c] = X[b, pad(s*ho + kh - pad), w, c]   (k/s x the input)
 *   zs_gemm_conv      : the existing implicit GEMM over W (taps along W, "batch" = (b, ho)) on the weight viewed as a Conv1d
 *                       weight [Cout, k*C, k];  weight gradient = zs_gemm_wgrad on the same view
 *   data gradient     : zs_gemm_conv(gather=1) into the W-padded domain, then zs_conv2d_fold (reflect folds along W, the
 *                       transpose of the H gather) -- a gather-formulated sum in fixed order, no atomics. */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx; int32_t x_f32;     /* [B][H_in][Wd] rows of ldx elements, C valid; fp32 or T */
  void* out; int64_t ldo;                         /* [B][H_out][Wd] rows of ldo >= k*C elements (T dtype; zeros past k*C) */
  int32_t B, H_in, H_out, Wd, C;
  int32_t k, stride, pad, pad_mode;
  /* full != 0: im2col along BOTH axes (the first layer, C = 1: the H-only gather pads its 5 columns per tap to a whole 128-byte
   * K chunk, 12.8 x the input; all 25 taps fit ONE chunk).  Output rows (b, ho, wo), wo < W_out = (Wd + 2 pad - k) / stride + 1,
   * columns (kh*k + kw)*C + c = x[b, pad(stride*ho + kh - pad), pad(stride*wo + kw - pad), c]; the convolution is then a
   * 1-tap zs_gemm_conv over these rows.  zs_conv2d_fold(full) is its transpose. */
  int32_t full;
} ZsConv2dGather;
int zs_conv2d_gather(const ZsConv2dGather* p, void* stream);
typedef struct {
  int32_t dtype;
  const void* gp; int64_t ldg;                    /* [B*H_out][Wd + 2*pad] rows, columns kh*C + c (zs_gemm_conv gather=1 output) */
  void* out; int64_t ldo; int32_t out_f32;        /* [B][H_in][Wd] rows; T dtype: zero filled to fill_cols */
  int32_t fill_cols;
  const void* add; int64_t ldadd;                 /* optional rows added (T dtype, same shape as out) */
  int32_t B, H_in, H_out, Wd, C;
  int32_t k, stride, pad, pad_mode;
  int32_t gp_rows;                                /* rows of gp per (b, h_out) block; 0 = Wd + 2*pad (a stride-2 data gradient is
                                                     computed by output parity and keeps an even number of rows per block) */
  int32_t full;                                   /* transpose of zs_conv2d_gather(full): gp rows (b, ho, wo), columns (kh*k + kw)*C + c */
} ZsConv2dFold;
int zs_conv2d_fold(const ZsConv2dFold* p, void* stream);
/* zs_conv1_fwd: the critic's FIRST Conv2d (one input channel, k x k taps, k*k <= 32, stride 2, padding k/2) computed directly from
 * the fp32 image: every lane of a wave builds its 8 bf16 im2col values (tap order kh*k + kw, as zs_conv2d_gather(full)) from the
 * image and one 16x16x32 MFMA per 16 output channels does the rest -- the [B*Ho*Wo][64] im2col buffer (12.8 x the image) is
 * neither written nor read:  out[(b, ho, wo)][co] = act(bias[co] + sum_{kh, kw} W[co][kh*k + kw] * bf16(x[b][refl(2ho + kh - k/2)][refl(2wo + kw - k/2)])).
 * bf16 only (the fp32 path keeps zs_conv2d_gather(full) + zs_gemm_conv); Cout a multiple of 16, <= 64 per launch column block. */
typedef struct {
  const float* x;                                 /* [B][H][W] */
  const void* W; int64_t ldw;                     /* bf16 [>= Cout][ldw], ldw >= 32, columns >= k*k are zeros (ConvLayer.wf of the layer) */
  const float* bias; int32_t act; float slope;    /* bias may be null; ZS_ACT_NONE / ZS_ACT_LRELU */
  void* out; int64_t ldo;                         /* bf16 [B*Ho*Wo][ldo] */
  int32_t B, H, Wd, Cout, k, pad_mode;
} ZsConv1Fwd;
int zs_conv1_fwd(const ZsConv1Fwd* p, void* stream);
/* zs_conv1_wgrad: weight and bias gradient of the same layer, again straight from the image (the im2col column of a position is
 * rebuilt in registers as the MFMA's B operand, the A operand is gz transposed):
 *   dW[co][kh*k + kw] (+)= sum_m gz[m][co] * bf16(x[...]),  db[co] (+)= sum_m gz[m][co]        (m over all B*Ho*Wo positions)
 * Positions are dealt to the waves of a fixed grid in a fixed order, each workgroup leaves one partial [64][32] in `workspace`,
 * a second kernel sums them in workgroup order: no atomics, bitwise reproducible.  bf16 gz; Cout a multiple of 16, <= 64. */
typedef struct {
  const float* x;                                 /* [B][H][W] */
  const void* gz; int64_t ldg;                    /* bf16 [B*Ho*Wo][ldg] */
  float* dW; int64_t lddw;                        /* fp32 [Cout][lddw], lddw >= k*k */
  float* db;                                      /* optional fp32 [Cout] */
  int32_t accumulate;
  int32_t B, H, Wd, Cout, k, pad_mode;
  float* workspace; size_t workspace_bytes;       /* >= zs_conv1_wgrad_workspace() bytes */
} ZsConv1Wgrad;
size_t zs_conv1_wgrad_workspace(void);
int zs_conv1_wgrad(const ZsConv1Wgrad* p, void* stream);
/* zs_conv2d_unpad: the last step of a stride-2 Conv2d's data gradient computed by output parity (ZsGemmConv with w_in > 0,
 * gather 1): class (ph, pw) holds the gradient at the positions (2 h2 + ph, 2 w2 + pw) of the PADDED input domain
 * [Hp][Wp] as rows [B][Hc(ph) * Wc(pw)], Hc(ph) = (Hp - ph + 1) / 2, Wc(pw) = (Wp - pw + 1) / 2.  This kernel removes the
 * padding -- reflect pads fold back onto the interior positions they mirrored (model/model.py pad_layer_2d), zero pads are
 * dropped -- and interleaves the classes:
 *   out[b][h*W + w][c] = sum over (h', w') in {(h, w) and its reflection partners} of g<(h'+pad)&1><(w'+pad)&1>[...] (+ add) */
typedef struct {
  int32_t dtype;
  const void* g00; const void* g01; const void* g10; const void* g11; int64_t ldg;      /* g<ph><pw> */
  int32_t B, H, W, C, Hp, Wp, pad, pad_mode;
  const void* add; int64_t ldadd;                 /* optional rows added (T dtype, same shape as out) */
  void* out; int64_t ldo; int32_t fill_cols;      /* [B][H*W] rows, zero filled to fill_cols */
} ZsConv2dUnpad;
int zs_conv2d_unpad(const ZsConv2dUnpad* p, void* stream);

/* zs_row_moments: per (b, c) sums over the T rows of a sample, two-stage in a fixed order (no atomics):
 *   u' = u - center_sum[b][c]*center_scale (if center_sum);  m = y ? lrelu'(y) : 1
 *   s1 = sum u'*m;  s2 = sum u'*m*v' (v' = u' when v == u; skipped if s2 == NULL);  s3 = sum u*w (if w)
 * Serves InstanceNorm2d forward (mean, then the centred second moment), its backward (sum g, sum g*a) and the three moments of
 * its double backward. */
typedef struct {
  int32_t dtype;
  const void* u; int64_t ldu;
  const void* v; int64_t ldv;
  const void* y; int64_t ldy; float slope;
  const void* w; int64_t ldw;
  const float* center_sum; float center_scale;
  int32_t B, T, C;
  float* s1; float* s2; float* s3;                /* [B][C] */
  float* partial; size_t partial_bytes;           /* >= zs_row_moments_workspace(B, T, C) */
} ZsRowMoments;
size_t zs_row_moments_workspace(int32_t B, int32_t T, int32_t C);
int zs_row_moments(const ZsRowMoments* p, void* stream);

/* zs_in2d_*: nn.InstanceNorm2d (no affine, biased variance, eps) + nn.Dropout2d over [B, T = H*W, C] rows.
 * dm[b][c] = keep / (1 - p) (1 when dropout is off); xhat = a * inv_dm (inv_dm = 1/dm, 0 for a dropped map).
 *  finalize : mean = s1/n, rstd = 1/sqrt(q/n + eps)         (s1 = sum y, q = sum (y - mean)^2 from zs_row_moments)
 *  fwd      : a = (y - mean) * rstd * dm
 *  bwd      : gz = [rstd*dm*(ga - S1/n) - rstd*xhat*(S2 + S2x)/n] * lrelu'(y)          (S1 = sum ga, S2 = sum ga*a)
 *  adj      : double backward of `bwd` for the gradient penalty: with gbar_y = gbar_z*lrelu'(y), A1 = sum gbar_y, A2 = sum gbar_y*a:
 *               gbar_a = dm*rstd*(gbar_y - A1/n - xhat*inv_dm*A2/n)                       (adjoint of ga)
 *               xbar_a = -inv_dm*rstd*(gbar_y*S2/n + ga*dm*inv_dm*A2/n)                   (adjoint of xhat, as a gradient w.r.t. a)
 *             the adjoint of rstd, A3 = sum gbar_z*gz, is added to S2 (S2x) in the following `bwd` through the forward graph. */
int zs_in2d_finalize(const float* s1, const float* q, float* mean, float* rstd, int64_t n_bc, int32_t T, float eps, void* stream);
/* zs_in2d_stats: mean and rstd of the forward pass in ONE pass over y (instead of zs_row_moments twice + finalize: the rows
 * of a 512-row slab are held in registers for a local mean and a local centred second moment; slabs are merged pairwise in a
 * fixed order with Chan's update  M2 = M2a + M2b + d^2 na nb / n, so the result is as accurate as the two-pass form).
 * y: [B][T] rows of ldy elements (T dtype), C % 8 == 0; partial: >= zs_row_moments_workspace(B, T, C) bytes. */
int zs_in2d_stats(int32_t dtype, const void* y, int64_t ldy, int32_t B, int32_t T, int32_t C, float eps, float* mean, float* rstd,
                  float* partial, size_t partial_bytes, void* stream);
typedef struct {
  int32_t dtype;
  const void* y; int64_t ldy; void* a; int64_t lda;
  const float* mean; const float* rstd; const float* dm;   /* dm may be NULL (1) */
  int32_t B, T, C;
} ZsIn2dFwd;
int zs_in2d_fwd(const ZsIn2dFwd* p, void* stream);
typedef struct {
  int32_t dtype;
  const void* ga; int64_t ldga; const void* ga2; int64_t ldga2;   /* ga2: optional second addend of the incoming gradient */
  const void* a; int64_t lda; const void* y; int64_t ldy;
  const float* S1; const float* S2; const float* S2x; const float* rstd; const float* dm;
  float slope;
  void* gz; int64_t ldgz;
  int32_t B, T, C;
} ZsIn2dBwd;
int zs_in2d_bwd(const ZsIn2dBwd* p, void* stream);
typedef struct {
  int32_t dtype;
  const void* gbz; int64_t ldgbz; const void* y; int64_t ldy; const void* a; int64_t lda; const void* ga; int64_t ldga;
  const float* A1; const float* A2; const float* S2; const float* rstd; const float* dm;
  float slope;
  void* gba; int64_t ldgba; void* xba; int64_t ldxba;
  int32_t B, T, C;
} ZsIn2dAdj;
int zs_in2d_adj(const ZsIn2dAdj* p, void* stream);

/* utils.calculate_gradients_penalty (utils.py:58-77) around the double backward:
 *  zs_lerp_rows : out[b] = alpha[b]*x[b] + (1 - alpha[b])*y[b]   (fp32 rows of n elements)
 *  zs_gp_penalty: s_b = sqrt(1e-12 + sum g_b^2);  *gp = mean_b (1 - s_b)^2;  gbar[b] = scale * (2/B) * (s_b - 1)/s_b * g[b] */
int zs_lerp_rows(const float* x, const float* y, const float* alpha, float* out, int32_t B, int64_t n, void* stream);
int zs_gp_penalty(const float* g, int32_t B, int64_t n, float scale, float* s_out, float* gp_out, float* gbar, void* stream);

/* Trainer.gen_step (trainer.py:266-278) and its gradient.  mode 0: x_gen = xd + m; 1 ('targeted_residual'): x_gen = xd + xd*m.
 * bwd: dpre = dx_gen * (mode ? xd : 1) * (tanh_out ? 1 - m^2 : m*(1 - m))   -- gradient w.r.t. the generator's pre-activation.
 * zs_l1_plain: loss = mean|a - b| (two-stage, fixed order), d = sign(a - b) * scale / n   (trainer.py:539 target-guided loss). */
/* xd, m: fp32 [rows][ld_in] (the decoders' output rows), x_gen / dx_gen: fp32 [rows][F] contiguous (the discriminator's input) */
int zs_gen_combine_fwd(const float* xd, const float* m, int64_t ld_in, float* x_gen, int64_t rows, int32_t F, int32_t mode, void* stream);
typedef struct {
  int32_t dtype;
  const float* dx_gen; int64_t ld_dx; const float* xd; const float* m; int64_t ld_in;
  void* dpre; int64_t ldo; int32_t fill_cols;                            /* T dtype rows */
  int64_t rows; int32_t F; int32_t mode, tanh_out;
} ZsGenCombineBwd;
int zs_gen_combine_bwd(const ZsGenCombineBwd* p, void* stream);
int zs_l1_plain(const float* a, const float* b, int64_t n, float scale, float* partial, float* loss_out, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
