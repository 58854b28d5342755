/*
 * jenga_amd.h -- C ABI of libjenga_amd.so: the MI355X (gfx950) AttenCarve hot path.
 *
 * The reference (dvlab-research/Jenga) is 100 % Python and has no FFI; its "native" code is one Triton
 * kernel plus calls into flash-attn / NCCL.  This header is the boundary a maintainer would bind instead
 * (ctypes stub: INTEGRATION.md, shipped binding: jenga_amd/_capi.py).  Every entry point names the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers; `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); every call only enqueues work on that stream, never synchronises, never allocates.
 *   - strides are in ELEMENTS; the innermost (head_dim) stride is always 1.
 *   - dtype: JENGA_BF16 or JENGA_FP16 (the reference kernel accepts both, ...diffres.py:167-170).
 *   - return value: JENGA_OK or an error code; jenga_last_error() gives the message for this thread.
 *   - the kernels work on 128-channel rows and 128-token blocks (the only configuration the reference runs: HunyuanVideo,
 *     HunyuanVideo-I2V, Wan2.1 all use D=128 and BLOCK_M=BLOCK_N=128).  The Triton kernel's other accepted head dims
 *     (16 / 32 / 64, ...diffres.py:155) are served by the host side on the same entry points: rows zero-padded to 128
 *     channels, the true head_dim ** -0.5 passed as sm_scale (jenga_bsattn_fwd) and as JENGA_SELECT_HEAD_DIM(d)
 *     (jenga_block_select) -- exact zeros in every dot product (jenga_amd/modules/attention_block_sparse.py).
 *   - state the library keeps (everything else is a pure function of its arguments):
 *       a thread-local error string (jenga_last_error);
 *       per device, for launches with JENGA_ATTN_BALANCE: 64 sets of 8 ticket counters in device memory, an event per
 *       set, a mutex and a hand-out cursor (csrc/lp_balance.h) -- a launch zeroes its set on its own stream, sets are
 *       reused behind their event, results do not depend on them;
 *       per process: JENGA_BALANCE_EXTRA_PCT read once at the first balanced launch; one "dynamic LDS size set" flag per
 *       kernel instantiation and device; the hipBLASLt handle, plan cache and timed algorithm choices of jenga_linear.
 */
#ifndef JENGA_AMD_H
#define JENGA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JENGA_ABI_VERSION 4   /* 2: jenga_block_select(flags), jenga_bsattn_fwd(order), jenga_sp_qkv_prologue, jenga_order_by_count, jenga_linear
                               * 3: jenga_sp_qkv_prologue takes (xq, xk) or xv alone; jenga_stream_delay; jenga_cross_attn_fwd;
                               *    jenga_linear_export_choices / _import_choices; jenga_linear refuses a workspace smaller than its plan's
                               * 4: jenga_pair_merge / jenga_bsattn_pair_fwd (the pair kernel) are part of the library; the experiment
                               *    launch flags (ping-pong 2, cohort 32, rotate 128) are gone */

enum { JENGA_OK = 0, JENGA_EINVAL = 1, JENGA_ELAUNCH = 2, JENGA_EUNSUPPORTED = 3 };
enum { JENGA_BF16 = 0, JENGA_FP16 = 1 };
#define JENGA_BLOCK 128
#define JENGA_HEAD_DIM 128

int jenga_abi_version(void);
const char* jenga_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Static geometry.  Replaces gilbert.py: gilbert_mapping :442-488 (sliced=0), sliced_gilbert_mapping
 * :332-440 (sliced=1): one thread per voxel evaluates the generalized-Hilbert index iteratively
 * (gilbert_xyz2d :12-38 / _r :68-272).  linear index = z*h*w + y*w + x.
 * l2h[linear] = curve index, h2l[curve index] = linear  (int64, length t*h*w each, device). */
int jenga_gilbert_map(void* stream, int t, int h, int w, int sliced, int64_t* l2h, int64_t* h2l);

/* Replaces gilbert_block_neighbor_mapping :597-677 and sliced_gilbert_block_neighbor_mapping :679-766:
 * out[nb*nb] bytes (0/1), nb = ceil(t*h*w / block); out[a][b] = 1 iff some voxel of curve-block a has a
 * 26-neighbour (or itself) in curve-block b.  `out` must be zero-filled by the caller. */
int jenga_gilbert_neighbors(void* stream, int t, int h, int w, int block, const int64_t* l2h, uint8_t* out);

/* ---------------------------------------------------------------------------------------------------
 * Hilbert gather / scatter.  Replaces `img[:, hilbert_order]` / `img[:, linear_to_hilbert]`
 * (jenga_hyvideo.py:116-118,226; jenga_hyvideo_multigpu.py:163-165,195; jenga_wan.py:559,655):
 * dst[b, i, :] = src[b, index[i], :] for i < n_rows; rows of row_bytes bytes (multiple of 16). */
int jenga_gather_rows(void* stream, const void* src, void* dst, const int64_t* index, int64_t batch,
                      int64_t n_rows, int64_t row_bytes, int64_t src_batch_stride_bytes,
                      int64_t dst_batch_stride_bytes);

/* ---------------------------------------------------------------------------------------------------
 * Fused per-head RMSNorm (+ optional RoPE) for one of Q/K.  Replaces RMSNorm.forward
 * (hyvideo/modules/norm_layers.py:43,56-59) followed by apply_rotary_emb's real branch
 * (hyvideo/modules/posemb_layers.py:206-212) as called at models_mul_block_gc_ha_multigpu.py:205-214,
 * 226-227, 420-435.  x/out: [B,S,H,128] with the given strides (out may alias x); weight: [128] in dtype
 * or NULL; cos/sin: [S_rope,128] fp32 or NULL (RoPE is applied to tokens s < s_rope only).
 * Rounding points are the reference's: fp32 normalise -> dtype -> * weight -> dtype -> fp32 rotate
 * (two products, one add, no FMA contraction) -> dtype.  eps < 0 skips the normalisation (RoPE only). */
int jenga_rmsnorm_rope(void* stream, const void* x, void* out, const void* weight, const float* cos,
                       const float* sin, int64_t B, int64_t S, int64_t H, int64_t x_sb, int64_t x_ss,
                       int64_t x_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, int64_t s_rope, float eps,
                       int dtype);

/* Wan flavour of the pre-ops (wan/modules/model_mul.py).
 * jenga_rmsnorm_rows: WanRMSNorm.forward :74-90 over the full model width C (1536 / 5120):
 *   out = (x.float()*rsqrt(mean(x^2)+eps)).type_as(x) * weight; weight_fp32 != 0: weight and out are fp32 (torch's
 *   promotion of dtype*fp32), else weight and out are in dtype.  x [rows, C] with a row stride.
 * jenga_rope_complex: rope_apply :40-71 -- adjacent pairs as complex numbers multiplied in float64 by
 *   cos/sin[s][64] (float64 tables already expanded per token, including the Hilbert freq_remap), rounded to fp32
 *   (out_dtype 2, what rope_apply returns) or on to bf16 (out_dtype 0, what the Wan attention op casts to,
 *   wan/modules/attention_block_triton_diffres.py:456-463).  in_dtype: 0 bf16, 1 fp16, 2 fp32.  Tokens >= s_rope are
 *   copied through (the reference concatenates x[i, seq_len:] back, :67). */
int jenga_rmsnorm_rows(void* stream, const void* x, void* out, const void* weight, int64_t rows, int64_t C,
                       int64_t x_row_stride, int64_t o_row_stride, float eps, int dtype, int weight_fp32);
int jenga_rope_complex(void* stream, const void* x, void* out, const double* cos, const double* sin, int64_t B,
                       int64_t S, int64_t H, int64_t x_sb, int64_t x_ss, int64_t x_sh, int64_t o_sb, int64_t o_ss,
                       int64_t o_sh, int64_t s_rope, int in_dtype, int out_dtype);
/* jenga_wan_norm_rope (round 3): the two above in ONE pass for the self-attention's q / k (model_mul.py:146-151 +
 *   the bf16 cast of the attention op): WanRMSNorm with its fp32 weight over the full width C, the float64 RoPE on rows
 *   < s_rope (tables [rows, 64], head_dim 128), one rounding to bf16 -- bit-identical to jenga_rmsnorm_rows
 *   (weight_fp32) followed by jenga_rope_complex (fp32 in, bf16 out), without the two fp32 tensors in between.  The output
 *   row stride lets the rows land in a buffer padded to a multiple of 128 tokens (the op's zero padding,
 *   wan/modules/attention_block_triton_diffres.py:448-451).  cos == sin == NULL: norm + cast only (cross-attention q).
 *   The ROW index is the RoPE position: `rows` are ONE sequence (batch 1, as WanSelfAttention runs it); the call is
 *   rejected when s_rope > rows. */
int jenga_wan_norm_rope(void* stream, const void* x, void* out, const float* weight, const double* cos,
                        const double* sin, int64_t rows, int64_t C, int64_t x_row_stride, int64_t o_row_stride,
                        int64_t s_rope, float eps, int dtype);

/* DiT block glue around the GEMMs: fused replacements of eager elementwise chains.
 * jenga_ln_modulate: modulate(LayerNorm(x), shift, scale) = LN(x)*(1+scale)+shift
 *   (models_mul_block_gc_ha_multigpu.py:196-199, 404; modulate_layers.py:31-50), LayerNorm without affine.
 *   mask != NULL selects (shift2, scale2) for rows with mask[row] != 0: the I2V "token_replace" modulation of the
 *   first-frame tokens (hyvideo_i2v/modules/modulate_layers.py:51-56).  shift/scale: [C] in dtype.
 * jenga_gate_residual: res + apply_gate(y, gate) (models_mul...:297-315, 500; modulate_layers.py:53-68), gate2/mask as
 *   above (hyvideo_i2v/modules/modulate_layers.py:84-95).
 * jenga_gelu_tanh: tanh-approximated GELU (mlp_act "gelu_tanh") between strided buffers. */
int jenga_ln_modulate(void* stream, const void* x, void* y, const void* shift, const void* scale, const void* shift2,
                      const void* scale2, const uint8_t* mask, int64_t rows, int64_t C, int64_t x_row_stride,
                      int64_t y_row_stride, float eps, int dtype);
int jenga_gate_residual(void* stream, const void* res, const void* y, const void* gate, const void* gate2,
                        const uint8_t* mask, void* out, int64_t rows, int64_t C, int64_t res_row_stride,
                        int64_t y_row_stride, int64_t o_row_stride, int dtype);
int jenga_gelu_tanh(void* stream, const void* x, void* out, int64_t rows, int64_t C, int64_t x_row_stride,
                    int64_t o_row_stride, int dtype);

/* Wan2.1 flavour of the same glue (wan/modules/model_mul.py:323-343): the residual stream x is fp32 (x + y*e is
 * evaluated under autocast(float32)), the inputs of the q/k/v, cross-attention and ffn GEMMs are its LayerNorm rounded
 * to the 16-bit dtype by autocast.
 * jenga_wan_ln_modulate: y = cast( LN(x) [*weight + bias] [*(1 + scale) + shift] )  -- norm1/norm2 with the (shift,
 *   scale) rows of `self.modulation + e` (:331-333, :339), norm3 with its affine (weight, bias) and no modulation
 *   (:338); all vectors fp32 [C]; either pair may be NULL; C % 4 == 0, C <= 6144.  round_ln != 0 rounds the
 *   LayerNorm result to the 16-bit dtype before the modulation: WanLayerNorm returns `.type_as(x)` (:99-104) and the
 *   first block receives the 16-bit patch embedding.
 * jenga_wan_gate_residual: out = x + float(y) [* gate]  (:334-335, :338, :340-341); out may alias x. */
int jenga_wan_ln_modulate(void* stream, const float* x, void* y, const float* weight, const float* bias,
                          const float* shift, const float* scale, int64_t rows, int64_t C, int64_t x_row_stride,
                          int64_t y_row_stride, float eps, int out_dtype, int round_ln);
int jenga_wan_gate_residual(void* stream, const float* x, const void* y, const float* gate, float* out, int64_t rows,
                            int64_t C, int64_t x_row_stride, int64_t y_row_stride, int64_t o_row_stride, int y_dtype);

/* jenga_linear: a dense layer of the DiT blocks with its element-wise neighbours in the
 * GEMM epilogue.  The GEMM is hipBLASLt's; this entry point only exposes what torch's front-end does not: an output
 * row stride, the GELU epilogue into a strided destination, and gate / residual as alpha-vector / C-matrix.
 *     out[m, n] = act( gate[n] * sum_k x[m,k] w[n,k]  +  bias[n]  +  res[m, n] )
 *   x [M,K], w [N,K] (nn.Linear layout), res / out [M,N], all 16-bit `dtype`, row strides in elements (inner stride 1);
 *   bias [N] in dtype (float32 with JENGA_BIAS_F32 in `act`) or NULL -- added AS GIVEN (when a gate is used the caller
 *   passes gate * bias, which is what apply_gate(linear(x)) means); gate fp32 [N] on the device or NULL (= 1); res NULL = no residual;
 *   act: JENGA_ACT_GELU_TANH (mlp_act "gelu_tanh"; not together with gate / res) or JENGA_ACT_NONE.
 *   workspace: device scratch the library may use (64 MiB is plenty); fp32 accumulation, ONE rounding to dtype at the
 *   end (the eager reference rounds after the GEMM, after the gate multiply and after the residual add).
 * Replaces: linear1's MLP half + nn.GELU(tanh) of MMSingleStreamBlock written straight into linear2's concat buffer
 * (models_mul_block_gc_ha_multigpu.py:404-406, 498-499); apply_gate + residual add behind img/txt_attn_proj, the MLPs'
 * fc2 and linear2 (:297-315, 500; modulate_layers.py:53-68). */
#define JENGA_ACT_NONE 0
#define JENGA_ACT_GELU_TANH 1
#define JENGA_BIAS_F32 256   /* OR-ed into `act` (ABI 3): bias is float32 [N] -- the gated bias gate * b stays unrounded */
/* OR-ed into `act` (ABI 4, round 6): res and out are FLOAT32 [M,N] (x, w stay 16-bit) -- the Wan blocks' fp32 residual stream,
 * `x = x + y * e` with y the 16-bit output of a linear layer (wan/modules/model_mul.py:334-341): out = res + gate * (x W^T) + bias
 * straight from the fp32 accumulator.  Where the reference rounds y to 16 bits before the multiply, this form does not (one
 * rounding to fp32 instead of three): closer to exact arithmetic, not bit-comparable -- jenga_wan_gate_residual stays for that.
 * Not together with an activation. */
#define JENGA_OUT_F32 512
int jenga_linear(void* stream, const void* x, const void* w, const void* bias, const void* res, const float* gate,
                 void* out, int64_t M, int64_t N, int64_t K, int64_t x_row_stride, int64_t w_row_stride,
                 int64_t res_row_stride, int64_t out_row_stride, int act, void* workspace, int64_t workspace_bytes,
                 int dtype);
/* Algorithm choices across the ranks of a job (ABI 3).  With JENGA_GEMM_CANDIDATES > 1 jenga_linear times the
 * heuristic's first candidates when it first meets a shape; ranks timing on their own may choose differently, and the
 * replicated text stream (models_mul_block_gc_ha_multigpu.py:196-214: every rank computes the same text rows) would then
 * no longer be bit-identical across ranks.  export: the calling device's plans as records of 12 int64 (M, N, K, x stride,
 * w stride, ldc, out stride, epilogue, mode, dtype, workspace bytes, the chosen algorithm -- packed, see below);
 * returns the number of plans (records beyond `capacity` are not written; records may be NULL to count).
 * import: those choices are used from now on for these shapes on every device of this process, without timing. */
int64_t jenga_linear_export_choices(int64_t* records, int64_t capacity);
int jenga_linear_import_choices(const int64_t* records, int64_t n);
/* (ABI 4) the last record field is packed: list index | (candidate count the list was requested with) << 8 | (hipBLASLt
 * solution index + 1) << 16.  The importer re-queries the heuristic with the EXPORTER's count (the list is not
 * prefix-stable across counts) and takes the algorithm with the exported solution index, the list position only as the
 * fallback.  jenga_linear_import_mismatches: plans rebuilt from an import so far whose solution index is not the exported
 * one (0 = every rank runs rank 0's algorithms). */
int64_t jenga_linear_import_mismatches(void);


/* jenga_qk_norm_rope_pool: jenga_rmsnorm_rope for Q AND K plus the two jenga_block_pool passes of a
 * layer in one kernel.  xq, xk [B, n_blocks*128, H, 128] share one set of strides (the q and k slices of a fused QKV
 * GEMM output), oq, ok likewise; RoPE on tokens < s_rope.  The mean of every 128-token block of the WRITTEN q / k rows
 * goes to qpool [B,H,nq_pool,128] / kpool [B,H,nk_pool,128] at block index pool_block0 + j (only where that index is
 * inside the pooled tensor; either pointer may be NULL) -- bit-identical to jenga_block_pool of the outputs, so that
 * image and text streams of a double block fill one pooled tensor in two calls.  H <= 64 (one thread group per head).
 * Replaces per layer: 2-4 jenga_rmsnorm_rope launches + 2 jenga_block_pool passes (1.4 GB of reads) and reads the
 * cos / sin tables once instead of twice. */
int jenga_qk_norm_rope_pool(void* stream, const void* xq, const void* xk, void* oq, void* ok, const void* wq,
                            const void* wk, const float* cosT, const float* sinT, void* qpool, void* kpool, int64_t B,
                            int64_t n_blocks, int64_t H, int64_t x_sb, int64_t x_ss, int64_t x_sh, int64_t o_sb,
                            int64_t o_ss, int64_t o_sh, int64_t s_rope, int64_t pool_block0, int64_t nq_pool,
                            int64_t nk_pool, float eps, int dtype);

/* jenga_sp_qkv_prologue (SURVEY.md §8 f-2 on the sequence-parallel path): the local half of the reference's chain
 * "RMSNorm(q), RMSNorm(k), apply_rotary_emb (models_mul_block_gc_ha_multigpu.py:196-214) -> SeqAllToAll4D's
 * scatter-heads permute of q, k and v (xdit_ring_atten.py:118-131)" in ONE pass: jenga_rmsnorm_rope of Q and K plus
 * jenga_ulysses_pack_heads of Q, K and V.  xq, xk, xv [B, S, H, 128] share one set of strides (the three slices of a
 * QKV GEMM output); any S.  Heads [head0, head0 + n_heads) are processed; head h is written to
 *     o + (h / heads_per_peer) * o_sp + b * o_sb + s * o_ss + (h % heads_per_peer) * o_sh        (elements)
 * for each of oq, ok (normalised, RoPE on tokens < s_rope) and ov (copied).
 *   peer-major send buffers [N][B][S][H/N][128]: head0 = 0, n_heads = H, o_sp = B*S*(H/N)*128;
 *   a rank's own head slice of the replicated text rows, written straight behind the gathered image rows of the
 *   attention inputs (xdit_ring_atten.py:159-175 slices them the same way): head0 = rank*H/N, n_heads = H/N, o_sp = 0.
 * Either the (xq, xk, oq, ok) group or (xv, ov) may be NULL (ABI 3): the sequence-parallel blocks issue the Q|K and
 * the V GEMM separately and post the Q, K exchange before the V GEMM runs (exchange / compute overlap).
 * Bit-identical to the unfused kernels. */
int jenga_sp_qkv_prologue(void* stream, const void* xq, const void* xk, const void* xv, void* oq, void* ok, void* ov,
                          const void* wq, const void* wk, const float* cosT, const float* sinT, int64_t B, int64_t S,
                          int64_t H, int64_t head0, int64_t n_heads, int64_t heads_per_peer, int64_t x_sb,
                          int64_t x_ss, int64_t x_sh, int64_t o_sp, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                          int64_t s_rope, float eps, int dtype);

/* ---------------------------------------------------------------------------------------------------
 * Block selection.  Replaces _build_block_index_with_importance_optimized
 * (hyvideo/modules/attention_block_triton_diffres.py:198-295; Wan first_frame_blocks rule
 * wan/modules/attention_block_triton_diffres.py:400-406).
 *
 * jenga_block_pool: mean over each 128-token block (:216-217). x [B,S,H,128] strided, n_blocks*128 <= S
 * tokens are pooled; pooled [B,H,n_blocks,128] contiguous in dtype (fp32 accumulate, one rounding). */
int jenga_block_pool(void* stream, const void* x, void* pooled, int64_t B, int64_t H, int64_t n_blocks,
                     int64_t x_sb, int64_t x_ss, int64_t x_sh, int dtype);

/* jenga_block_select: scores = dtype(dtype(qpool.kpool^T) * 128^-0.5) (:221-232); softmax over the
 * first nk_img columns in fp32 -> dtype (:235-238); descending sort (ties: lower index first; the
 * reference's sort is unstable, any tie order is legal there); cumulative sum accumulated sequentially in
 * fp32 with each partial rounded to dtype, compared with p rounded to dtype (:241-247);
 * n = max(#(cumsum <= p) + 1, top_k) (:245-250); keep the first n sorted columns (:253-276); OR the static
 * neighbour rows (:280-289); first_frame rule (Wan); all text columns [nk_img, nk_img+text_blocks) (:292-293).
 * CONTRACT of the cumulative sum: the parity target is what torch.cumsum computes for a 16-bit tensor on the CPU
 * (one fp32 running sum per row, every partial rounded to dtype) -- the semantics of the reference's goldens, which
 * can only be generated on the CPU here.  torch's DEVICE cumsum of a bf16 row is a blocked scan whose partials round
 * differently; against it the kept count of a 900-block row differs by up to 3 / 7 / 18 blocks at p = 0.3 / 0.5 /
 * 0.9 on flat scores (tests/test_gpu_select.py records the measured difference,
 * profiles/r02_parity_cumsum_semantics.json) and by 0 where top_k or a peaked softmax decides n.
 *   qpool [B,H,nq,128], kpool [B,H,nk_all,128] (from jenga_block_pool), nk_all = nk_img + text_blocks
 *   neighbors: uint8 [nb_rows, nb_cols] row-major (row stride nb_cols) or NULL
 * Outputs (either may be NULL):
 *   mask  uint8 [B,H,nq,nk_all]            the reference's one-hot layout (for parity checks)
 *   idx   int32 [B,H,nq,nk_all], cnt int32 [B,H,nq]   ascending kept-column lists for jenga_bsattn_fwd
 * flags: JENGA_SELECT_DEVICE_SCAN replaces the cumulative sum by what torch.cumsum computes for a 16-bit tensor ON THE
 *   DEVICE (the reference as shipped runs :241-250 on a CUDA bf16 tensor): ATen's blocked scan
 *   (ATen/native/cuda/ScanUtils.cuh, torch 2.10: tensor_kernel_scan_innermost_dim_impl) restated -- chunks of
 *   2 * 2^l columns, l = get_log_num_threads_x_inner_scan(B*H*nq, nk_img) (16 columns x 2 at the production shapes), a
 *   Sklansky network per chunk with EVERY add rounded to the 16-bit dtype, the chunk total carried in the 16-bit dtype;
 *   all columns with cumsum <= p are counted.  tests/test_gpu_select.py checks the kept counts against torch.cumsum on
 *   the device bit for bit.  Default (0): the CPU semantics above.
 *   JENGA_SELECT_HEAD_DIM(d), d in {16, 32, 64}: the scores are scaled by float(d ** -0.5) instead of float(128 ** -0.5)
 *   (:232 `* head_dim ** -0.5`) -- for heads narrower than 128 channels, whose pooled rows the caller hands over
 *   zero-padded to 128 (the Triton kernel accepts head dims 16 / 32 / 64 / 128, :155; zero channels add exact zeros to
 *   every dot product).  Bits 8..15 of flags; 0 = 128. */
#define JENGA_SELECT_DEVICE_SCAN 1
#define JENGA_SELECT_HEAD_DIM(d) (((d) & 0xFF) << 8)
int jenga_block_select(void* stream, const void* qpool, const void* kpool, const uint8_t* neighbors,
                       int64_t nb_rows, int64_t nb_cols, uint8_t* mask, int32_t* idx, int32_t* cnt,
                       int64_t B, int64_t H, int64_t nq, int64_t nk_img, int64_t text_blocks, int64_t top_k,
                       float p, int64_t first_frame_blocks, int dtype, int flags);

/* ---------------------------------------------------------------------------------------------------
 * Block-sparse attention forward.  Replaces _triton_block_sparse_attn_fwd_kernel_onehot + launcher
 * (hyvideo/modules/attention_block_triton_diffres.py:38-136, 139-196) for the image query blocks and the
 * flash_attn_func call for the text query blocks (:371-380), in ONE launch.
 *
 * Step 1, jenga_pack_v: re-tiles V into the MFMA operand order used by the P.V product
 *   vt workspace bytes = B*H*dst_blocks_total*128*128*2.  v [B, n_blocks*128, H, 128] strided supplies kv blocks
 *   [dst_block0, dst_block0 + n_blocks) of the workspace, so image and text V (separate GEMM outputs in the
 *   double-stream blocks) are packed without a torch.cat.
 * Step 2, jenga_bsattn_fwd:
 *   q,k,o [B,S,H,128] strided (S = n_blocks*128), vt from step 1, seqlens int32 [B] (device),
 *   image query blocks m < nq_img use idx/cnt (layout of jenga_block_select, row stride = n_blocks):
 *       q~ = dtype(q * (sm_scale*log2 e)); s = q~.k^T (fp32) [+ text_amp if kv block >= text_block_start];
 *       kv columns >= seqlens[b] masked; base-2 online softmax; P rounded to dtype before P.V; o = acc/l.
 *       Rows >= seqlens[b] are written as zeros (the reference leaves its pre-zeroed output untouched).
 *   text query blocks m >= nq_img attend to every kv block with plain softmax(q.k^T*sm_scale), no length
 *   mask and no text_amp (flash_attn_func semantics).
 *   idx/cnt may be NULL when nq_img == 0 (fully dense: sa_drop_rate == 0 for a single segment). */
size_t jenga_pack_v_bytes(int64_t B, int64_t H, int64_t n_blocks);
int jenga_pack_v(void* stream, const void* v, void* vt, int64_t B, int64_t H, int64_t n_blocks, int64_t v_sb,
                 int64_t v_ss, int64_t v_sh, int64_t dst_block0, int64_t dst_blocks_total, int dtype);
int jenga_bsattn_fwd(void* stream, const void* q, const void* k, const void* vt, void* o,
                     const int32_t* seqlens, const int32_t* idx, const int32_t* cnt, const int32_t* order, int64_t B,
                     int64_t H, int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                     int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                     float sm_scale, float text_amp, int64_t text_block_start, int dtype, int flags);
/* flags: launch order and which of the parity-equivalent kernels runs (DESIGN.md section 3 has the measurements)
 *   no LP bit: the round-1 kernel (csrc/bsattn.hip), 4-wave workgroup per 128-row query block */
#define JENGA_ATTN_XCD_REMAP 1 /* contiguous q-block ranges per XCD (L2 locality); 0 = plain head-major order */
#define JENGA_ATTN_BALANCE 4   /* (LP and pair kernels with XCD_REMAP; part of the Python modules' default) every workgroup
                                  DRAWS its query block (pair): a ticket from the queue of the XCD it runs on (the same
                                  contiguous range and order as the static mapping) and, once that is empty, from the
                                  fullest other queue; the grid is oversubscribed by 1/8.  Evens out the speed differences
                                  between the 8 XCDs of a chip (3-8 % in workgroup lifetime).  Bit-identical results.  Every
                                  launch in flight has its own ticket counters (64 sets per device, reused behind an event);
                                  a capturing stream gets the static mapping */
#define JENGA_ATTN_LP 8        /* same decomposition, in-wave software pipeline (softmax inside the MFMA stream):
                                  csrc/bsattn3.hip */
/* (bits 2, 32 and 128 were the ping-pong, cohort and rotated-walk experiments of rounds 1-4: measured, recorded in
 * DESIGN.md / profiles/, removed in ABI version 4; they are ignored) */
/* order (may be NULL): int32 [B,H,nq_img], launch position -> image query block, a permutation per (b, h) -- a
 *   scheduling hint only (every query block is computed exactly once either way, results are bit-identical).
 *   jenga_order_by_count fills it from cnt: inside every segment of `segment` consecutive query blocks the blocks are
 *   ordered by DESCENDING kept count (ties: lower block first), so that the longest lists of a range start first and
 *   the short ones fill the tail (reference grid: attention_block_triton_diffres.py:165 launches in plain order).
 *   With JENGA_ATTN_XCD_REMAP use segment = ceil(nq_img / 8) (= one XCD's contiguous range); without it
 *   segment = nq_img.  The LP kernel honours it; the round-1 kernel ignores it. */
int jenga_order_by_count(void* stream, const int32_t* cnt, int64_t BH, int64_t nq_img, int64_t segment,
                         int32_t* order);

/* ---------------------------------------------------------------------------------------------------
 * Block-sparse attention forward on query-block PAIRS (csrc/bsattn5.hip, round 5; same reference lines and semantics as
 * jenga_bsattn_fwd).  Two Hilbert-adjacent query blocks per workgroup, one wave per SIMD, 64 query rows per wave: a kv
 * block kept by BOTH is staged once for 256 query rows and every K / V^T fragment read from LDS feeds two MFMAs.
 *   jenga_pair_merge: idx/cnt of jenga_block_select ->
 *       pidx int32 [B,H,ceil(nq_img/2),n_blocks]: per query-block pair (2j, 2j+1) the kv blocks both rows keep, then
 *            those only row 2j keeps, then those only row 2j+1 keeps -- each part ascending;
 *       pcnt int32 [B,H,ceil(nq_img/2),4]: the three part lengths, 0.   (odd nq_img: the last pair has one row)
 *   jenga_bsattn_pair_fwd: same arguments and semantics as jenga_bsattn_fwd with (pidx, pcnt) in place of (idx, cnt);
 *       text query blocks run as pairs of their own in the same launch.  The kv blocks of a row are visited in the order
 *       (only-this-row, shared), each part ascending, instead of ascending overall: online softmax is order independent up
 *       to fp32 rounding of the running sums and every rescale stays an exact power of two -- results are deterministic and
 *       equal to jenga_bsattn_fwd's within that rounding, not bit for bit.
 *   order (may be NULL): int32 [B,H,ceil(nq_img/2)], launch position -> pair, a permutation per (b, h): the scheduling
 *       hint of jenga_bsattn_fwd for pairs (jenga_order_by_count on the pairs' work 2 n_shared + n_a + n_b, segment =
 *       ceil(npairs / 8) with JENGA_ATTN_XCD_REMAP); results do not depend on it.
 *   flags: JENGA_ATTN_XCD_REMAP (contiguous pair ranges per XCD), JENGA_ATTN_BALANCE (pairs drawn from per-XCD queues). */
int jenga_pair_merge(void* stream, const int32_t* idx, const int32_t* cnt, int64_t B, int64_t H, int64_t nq_img,
                     int64_t n_blocks, int32_t* pidx, int32_t* pcnt);
int jenga_bsattn_pair_fwd(void* stream, const void* q, const void* k, const void* vt, void* o,
                          const int32_t* seqlens, const int32_t* pidx, const int32_t* pcnt, const int32_t* order,
                          int64_t B, int64_t H, int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                          int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float sm_scale,
                          float text_amp, int64_t text_block_start, int dtype, int flags);

/* ---------------------------------------------------------------------------------------------------
 * Dense cross-attention (ABI 3).  Replaces the flash_attention call of WanT2VCrossAttention.forward
 * (wan/modules/model_mul.py:183-205; wan/modules/attention.py flash_attention with k_lens = None): every query
 * row attends to ALL nkv_blocks*128 keys, softmax(q.k^T * sm_scale) in fp32, P rounded to dtype before P.V.
 * The same kernel as the text rows of jenga_bsattn_fwd (LP kernel, TEXT mode), with a kv sequence of its own
 * length.  q [B, nq_blocks*128, H, 128] strided (pad the last block; padded rows produce padded output rows),
 * k [B, nkv_blocks*128, H, 128] strided, vt = jenga_pack_v(v, n_blocks = nkv_blocks), o like q.  Keys >= kv_len
 * (kv_len inside the last kv block: the buffers are padded to whole blocks, their contents there are ignored) are
 * masked out -- the reference's 512-token context needs none, a shorter context does. */
int jenga_cross_attn_fwd(void* stream, const void* q, const void* k, const void* vt, void* o, int64_t B, int64_t H,
                         int64_t nq_blocks, int64_t nkv_blocks, int64_t kv_len, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                         int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                         float sm_scale, int dtype);

/* ---------------------------------------------------------------------------------------------------
 * Ulysses head pack/unpack: the local halves of xFuserLongContextAttention.forward's SeqAllToAll4D calls
 * (hyvideo/modules/xdit_ring_atten.py:118-131 scatter heads / gather sequence, :212-217 the reverse).
 * The exchange itself is one RCCL all_to_all_single issued from the host (torch.distributed); these
 * kernels produce / consume its peer-major buffers, replacing yunchang's permute+contiguous copies.
 *   pack:   x [B, S_loc, H, 128] strided -> send [N][B][S_loc][H/N][128]: chunk r holds heads
 *           [r*H/N, (r+1)*H/N) and goes to rank r.  After the all-to-all, chunk r of the receive buffer is
 *           rank r's S_loc tokens for MY heads, i.e. for B == 1 the receive buffer already IS the gathered
 *           [1, N*S_loc, H/N, 128] tensor (rank-major sequence) and needs no unpack.
 *   unpack: recv [N][B][S_loc][H/N][128] -> y [B, S_loc, H, 128] strided (inverse map; used after the
 *           reverse all-to-all, whose send buffer for B == 1 is the attention output as it stands). */
int jenga_ulysses_pack_heads(void* stream, const void* x, void* send, int64_t B, int64_t S_loc, int64_t H,
                             int64_t N, int64_t x_sb, int64_t x_ss, int64_t x_sh);
int jenga_ulysses_unpack_heads(void* stream, const void* recv, void* y, int64_t B, int64_t S_loc, int64_t H,
                               int64_t N, int64_t y_sb, int64_t y_ss, int64_t y_sh);

/* ---------------------------------------------------------------------------------------------------
 * Measurement aid, not part of the reference's path: one wavefront keeps `stream` busy for `microseconds` of the
 * device's constant-rate wall clock.  bench.py --simulate-ranks N puts it on a side stream in front of the local
 * copies that stand in for the Ulysses exchanges (xdit_ring_atten.py:118-131, 212-217), so that the overlap of
 * the exchanges with the blocks' GEMMs can be timed on ONE GPU at a stated xGMI rate. */
int jenga_stream_delay(void* stream, double microseconds);

#ifdef __cplusplus
}
#endif
#endif /* JENGA_AMD_H */
