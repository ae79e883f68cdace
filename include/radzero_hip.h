/* radzero_hip.h — C-ABI of libradzero_hip.so: MI355X (gfx950) implementation of RadZero's VL-CABS
 * zero-shot inference hot path.
 *
 * The reference (deepnoid-ai/RadZero) has NO native/FFI layer: its boundary is the Python method
 * protocol of `CxrAlignModel` (exp/cxr_pt/model/modeling.py).  Each entry point below names the
 * reference interface it replaces; INTEGRATION.md shows the ctypes binding a reference maintainer
 * would add.  Conventions:
 *   - plain pointers and sizes only (no torch types); every *_dev pointer is device memory owned by the
 *     caller (e.g. a torch tensor's data_ptr()); every *_host pointer is host memory;
 *   - every function returns 0 on success, otherwise an rz_status / hipError_t code; the message of the
 *     last failure on the calling thread is available from rz_last_error();
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); calls are
 *     asynchronous on that stream; a handle is not re-entrant (one handle per GPU / process);
 *   - the library allocates only packed weights + workspaces (rz_load_weight / rz_reserve), never in
 *     the forward calls (hipGraph-capturable).
 */
#ifndef RADZERO_HIP_H
#define RADZERO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rz_model* rz_handle_t;

enum rz_dtype { RZ_F32 = 0, RZ_BF16 = 1, RZ_F16 = 2 };

enum rz_status {
    RZ_OK = 0,
    RZ_ERR_INVALID = 10001,      /* bad argument / shape (reference raises ValueError) */
    RZ_ERR_STATE = 10002,        /* weights / workspace / position table not ready */
    RZ_ERR_UNSUPPORTED = 10003,  /* reference raises NotImplementedError (modeling.py:108, :206) */
    RZ_ERR_NOMEM = 10004
};

/* Hyper-parameters.  Field names follow exp/cxr_pt/configs/radzero.yaml:16-46,
 * exp/cxr_pt/model/configuration.py:91-104 and the HF Dinov2Config / MPNetConfig the reference builds. */
typedef struct rz_config {
    int32_t compute_dtype;            /* rz_dtype of MFMA operands; statistics/residual/VL-CABS stay fp32 */
    int32_t hidden_size;              /* 768 */
    int32_t num_attention_heads;      /* 12 */
    int32_t mlp_ratio;                /* 4 */
    int32_t patch_size;               /* 14 */
    int32_t num_channels;             /* 3 */
    int32_t vit_layers;               /* 12: transformers Dinov2Model (vision_encoders.py:28-29) */
    int32_t align_layers;             /* 2:  AlignTransformer (align_transformers.py:23-45), no final LN */
    float vit_layer_norm_eps;         /* 1e-6 */
    int32_t vocab_size;               /* 30527 */
    int32_t max_position_embeddings;  /* 514 */
    int32_t text_layers;              /* 12: MPNetModel (text_encoders.py:13-14) */
    int32_t text_intermediate_size;   /* 3072 */
    float text_layer_norm_eps;        /* 1e-5 */
    int32_t pad_token_id;             /* 1 */
    float shared_layer_norm_eps;      /* 1e-5: RadZeroLoss.layer_norm (losses.py:51) */
} rz_config;

const char* rz_last_error(void);
const char* rz_version(void);

/* ---- lifecycle: replaces CxrAlignModel.__init__ + from_pretrained (modeling.py:51-94, README.md:77-82) ---- */
int rz_create(const rz_config* cfg, rz_handle_t* out);
int rz_destroy(rz_handle_t h);

/* One tensor of the checkpoint, by its state_dict name (modeling.py:55-86; e.g.
 * "vision_model.encoder.layer.3.mlp.fc1.weight", "align_transformer.transformer_layers.layer.0.norm1.bias",
 * "text_model.encoder.layer.0.attention.attn.q.weight", "loss_fns.RadZeroLoss.loss_temperature").
 * fp32 host data; packed / cast / fused (q scale 1/sqrt(dh) folded) on upload.  Unknown names that the path
 * does not use (mask_token, pooler) are accepted and ignored (returns 0). */
int rz_load_weight(rz_handle_t h, const char* name, const float* data_host, int64_t numel);
/* 0 when every tensor the path needs has been loaded; otherwise RZ_ERR_STATE and rz_last_error() names one. */
int rz_weights_ready(rz_handle_t h);

/* Position table for one patch grid: pos_host is (1 + grid_h*grid_w, hidden) fp32, the output of
 * Dinov2Embeddings.interpolate_pos_encoding (bicubic, once per resolution, host side). */
int rz_set_position_table(rz_handle_t h, int grid_h, int grid_w, const float* pos_host);

/* Workspace for up to max_batch images of max_tokens (= 1 + grid_h*grid_w) tokens, max_prompts prompts of
 * max_prompt_len tokens.  Re-callable (grows only). */
int rz_reserve(rz_handle_t h, int max_batch, int max_tokens, int max_prompts, int max_prompt_len);
/* Token rows per image that rz_vision_forward gives a batch of `batch` images of n_tokens (= 1 + grid_h*grid_w) tokens under the handle's
 * "pad_rows" rule (batch = 0: the batch-independent upper bound that sizes tables and workspaces).  Host arithmetic only; it lets a batch
 * driver price a batch's GEMM tile rounds (radzero_amd/inference.py batch shaping; the reference's drivers take the DataLoader's batch as
 * it comes: exp/cxr_pt/inference/utils.py:81-100) without re-implementing the rule. */
int rz_padded_tokens(rz_handle_t h, int n_tokens, int batch, int* n_pad_out);

/* ---- CxrAlignModel.forward_vision_model (modeling.py:96-123): Dinov2Model + AlignTransformer ----
 * pixel_values_dev: fp32 (batch, channels, height, width).  The aligned tokens stay inside the handle
 * (input of rz_vlcabs); vision_tokens_out_dev, if not NULL, receives them as fp32 (batch, N, hidden). */
int rz_vision_forward(rz_handle_t h, const float* pixel_values_dev, int batch, int channels, int height, int width,
                      float* vision_tokens_out_dev, void* stream);

/* ---- CxrAlignModel.forward_text_model, MPNet branch (modeling.py:128-156): encoder + masked mean pool ----
 * input_ids_dev / attention_mask_dev: int64 (n_prompts, len).  rel_bias_dev: fp32 (heads, len, len) =
 * relative_attention_bias[bucket(j - i)] (MPNetEncoder.compute_position_bias, computed once by the host).
 * text_features_out_dev: fp32 (n_prompts, hidden) = "text_features_wo_l2_norm".
 * fp32 mode (round 6): the GEMMs run on the three-plane f16 form (option "gemm_f32_split"; 2.5e-5 from the exact kernels' result) with a predicated
 * overflow guard of their own (option "f32_split_guard": a prompt encode never trips the vision forward's guard; its repeats are counted into the
 * same re-run counter that rz_get_model_option reports). */
int rz_text_forward(rz_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev, int n_prompts,
                    int len, const float* rel_bias_dev, float* text_features_out_dev, void* stream);

/* ---- RadZeroLoss.forward(compute_loss=False) + SimilarityLogit + final scaling
 *      (losses.py:71-105, :187-240; modeling.py:311-328) on the tokens of the last rz_vision_forward ----
 * text_features_dev: fp32 (n_prompts, hidden), pre-LayerNorm.
 * scores_out_dev: fp32 (batch, n_prompts, N) = "t2i_attn_weights"[0] (cosine / tau, CLS column included;
 *                 similarity_scores is its [:, :, 1:] view).
 * t2i_logits_out_dev: fp32 (n_prompts, batch).  logits_out_dev: fp32 (batch, n_prompts) = t2i^T / tau. */
int rz_vlcabs(rz_handle_t h, const float* text_features_dev, int n_prompts, int batch, float* scores_out_dev,
              float* t2i_logits_out_dev, float* logits_out_dev, void* stream);

/* ---- interpolate_similarity_scores, BlipImageProcessor branch (inference/segmentation_utils.py:62-70)
 *      (+ torch.sigmoid of attention_map_base.py:57 when apply_sigmoid != 0) ----
 * maps_dev: n_maps maps of grid*grid fp32, consecutive maps map_stride floats apart; out_dev (n_maps, H, W). */
int rz_upsample_maps(rz_handle_t h, const float* maps_dev, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                     int apply_sigmoid, float* out_dev, void* stream);

/* AspectRatioBlipImageProcessor branch of the same helper (segmentation_utils.py:41-60): keep_aspect_ratio != 0 upsamples to the
 * padded square max(H, W) and returns the crop [pad_top : pad_top + H, pad_left : pad_left + W] (never materialising the square). */
int rz_upsample_maps_ex(rz_handle_t h, const float* maps_dev, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                        int apply_sigmoid, int keep_aspect_ratio, float* out_dev, void* stream);

/* ---- get_grounding_point, BlipImageProcessor branch (inference/grounding_utils.py:166-261): flat argmax of the
 *      bilinear-upsampled map -> (x, y), fused (the n_maps x H x W map is never written: 4.3 GB at BASELINE cfg 4) ----
 * xy_out_dev: int32 (n_maps, 2) = (x_index, y_index); keys_ws_dev: scratch of n_maps x 8 bytes. */
int rz_grounding_points(rz_handle_t h, const float* maps_dev, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                        int32_t* xy_out_dev, void* keys_ws_dev, void* stream);

/* same with the AspectRatioBlipImageProcessor branch (grounding_utils.py:172-190) when keep_aspect_ratio != 0 */
int rz_grounding_points_ex(rz_handle_t h, const float* maps_dev, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                           int keep_aspect_ratio, int32_t* xy_out_dev, void* keys_ws_dev, void* stream);

/* ---- InferDataset collate_fn + Blip image processor (inference/dataset.py:31-51; processing.py:31-49, :90-91) on the device:
 *      cv2.normalize(NORM_MINMAX) to 8 bit (optional) -> grey->RGB -> Pillow bicubic resize of the uint8 image (same 22-bit
 *      fixed-point tables, built by the host: radzero_amd/preprocess.py) -> * rescale -> (x - mean) / std ----
 * image_dev: (height, width, channels) of src_dtype 0 = uint8, 1 = uint16, 2 = float32; channels 1 or 3.
 * bounds_*_dev int32 (out, 2) = (first input index, count); coeffs_*_dev int32 (out, ksize).
 * workspace_dev: 16 + H*W*C + H*S*C + S*S*C bytes.  pixel_values_out_dev: fp32 (3, out_side, out_side). */
int rz_preprocess_image(const void* image_dev, int src_dtype, int height, int width, int channels, int out_side,
                        const int32_t* bounds_h_dev, const int32_t* coeffs_h_dev, int ksize_h, const int32_t* bounds_v_dev,
                        const int32_t* coeffs_v_dev, int ksize_v, const float* mean3_host, const float* std3_host, float rescale,
                        int minmax_normalize, void* workspace_dev, float* pixel_values_out_dev, void* stream);

/* The same for a BATCH of images of different sizes / dtypes in one set of launches (collate_fn processes the whole batch,
 * dataset.py:31-51), with the AspectRatioBlipImageProcessor branch (processing.py:232-259: convert_to_rgb -> pad_to_square with
 * fill 0 -> BlipImageProcessor) when pad_left / pad_top / padded_* describe the padded square:
 *   pad_left = (max(w, h) - w) / 2, pad_top = (max(w, h) - h) / 2, padded_height = padded_width = max(w, h);
 * for the plain BlipImageProcessor pad_* = 0 and padded_* = height / width.  The resampling tables are those of
 * padded_width -> out_side (horizontal) and padded_height -> out_side (vertical).  descs_host: host array, copied before the call returns (as kernel
 * arguments: the host never waits for the stream, and the call may be captured into a hipGraph). */
typedef struct rz_image_desc {
    const void* image_dev;            /* (height, width, channels) of src_dtype: 0 uint8, 1 uint16, 2 float32 */
    int32_t src_dtype, height, width, channels;
    int32_t pad_left, pad_top, padded_height, padded_width;
    const int32_t* bounds_h_dev; const int32_t* coeffs_h_dev; int32_t ksize_h;
    const int32_t* bounds_v_dev; const int32_t* coeffs_v_dev; int32_t ksize_v;
} rz_image_desc;
/* bytes of workspace_dev the call below needs for these images (0 on a bad argument) */
size_t rz_preprocess_batch_workspace(const rz_image_desc* descs_host, int n_images, int out_side);
/* pixel_values_out_dev: fp32 (n_images, 3, out_side, out_side) — directly the input of rz_vision_forward */
int rz_preprocess_batch(const rz_image_desc* descs_host, int n_images, int out_side, const float* mean3_host, const float* std3_host,
                        float rescale, int minmax_normalize, void* workspace_dev, size_t workspace_bytes, float* pixel_values_out_dev,
                        void* stream);

/* ---- per-kernel entry points (used by the parity tests; all pointers device) ---- */
/* C = A[M,K] W[N,K]^T + bias; dtype of A/W/out = rz_dtype; epilogue: 0 store, 1 GELU(erf), 7 store fp32 */
int rz_gemm(int dtype, int epilogue, const void* a_dev, const void* w_dev, const float* bias_dev, void* out_dev, int m,
            int n, int k, void* stream);
/* every fused epilogue of the GEMM (radzero_amd/csrc/rz_kernels.h `Epilogue`): 0 store, 1 GELU, 2 per-head q|k layout,
 * 3 transposed per-head v layout, 4 resid += scale*(acc+bias), 5 out_f32 = acc+bias+resid, 6 patch-embed table add,
 * 7 store fp32.  Leading dimensions in elements. */
int rz_gemm_ex(int dtype, int epilogue, const void* a_dev, int64_t lda, const void* w_dev, int64_t ldw, const float* bias_dev,
               void* out_dev, int64_t ldo, const float* scale_dev, float* resid_dev, int64_t ldr, int rows_per_image,
               int heads_total, int m, int n, int k, void* stream);
/* The block's q|k|v projection as the model launches it (TF:dinov2/modeling_dinov2.py:199-213; replaces three nn.Linear):
 * x[M,K] times w_qkv[3K,K]^T (rows q | k | v) + bias[3K]  ->  qk_out (B, 2*heads, rows_per_image, 64) and
 * v_t_out (B, heads, 64, rows_per_image), K = heads*64.  ONE launch where the persistent 256x256 kernel applies
 * (16-bit dtype, M % 256 == 0, enough tiles), otherwise the q|k and v launches; *fused_out (may be NULL) reports which. */
int rz_gemm_qkv(int dtype, const void* x_dev, const void* w_qkv_dev, const float* bias_dev, void* qk_out_dev, void* v_t_out_dev,
                int rows_per_image, int heads, int m, int* fused_out, void* stream);
/* LayerNorm rows of `dim` (=768) fp32 -> out_t_dev (dtype, may be NULL) and/or out_f32_dev (may alias in) */
int rz_layernorm(int dtype, const float* in_dev, const float* gamma_dev, const float* beta_dev, float eps, void* out_t_dev,
                 float* out_f32_dev, int64_t rows, int dim, void* stream);
/* flash attention, heads of 64: q,k (B,H,n_pad,64), v_t (B,H,64,n_pad), ctx (B*n_pad, H*64).  Scores are taken in
 * log2 units: ctx = softmax2(q k^T) v with softmax2(x) = 2^x / sum 2^x  (the model folds log2(e)/sqrt(64) into q). */
int rz_flash_attention(int dtype, const void* q_dev, const void* k_dev, const void* v_t_dev, void* ctx_dev, int batch,
                       int heads, int n_valid, int n_pad, void* stream);
/* the same contraction for fp32 tensors on the f16 matrix pipe: every operand carried as hi + lo f16 planes (22 mantissa bits),
 * every product as three MFMAs with fp32 accumulation (what the fp32 mode's attention runs by default, option "attn_f32_split").
 * workspace_dev: rz_flash_attention_split_workspace(batch, heads, n_pad) bytes, caller owned. */
size_t rz_flash_attention_split_workspace(int batch, int heads, int n_pad);
int rz_flash_attention_f32_split(const float* q_dev, const float* k_dev, const float* v_t_dev, float* ctx_dev, void* workspace_dev,
                                 int batch, int heads, int n_valid, int n_pad, void* stream);
/* the same with the two correction terms of every product as ONE block-scaled e4m3 MFMA (the "MX" form the fp32 mode runs wherever
 * a launch's rows are a multiple of 256, option "gemm_f32_mx"): hi f16 planes + e4m3 pair planes are built in the same workspace */
int rz_flash_attention_f32_mx(const float* q_dev, const float* k_dev, const float* v_t_dev, float* ctx_dev, void* workspace_dev,
                              int batch, int heads, int n_valid, int n_pad, void* stream);

/* ---- the text side and the patch embedding, kernel by kernel (the model reaches them through rz_text_forward /
 *      rz_vision_forward) ---- */
/* MPNetEmbeddings (TF:mpnet/modeling_mpnet.py:58-95; position ids skip pads, :873-881): word_emb[ids] + pos_emb[pos_ids] -> LayerNorm.
 * input_ids_dev int64 (n_prompts, len); word_emb_dev fp32 (vocab, 768); pos_emb_dev fp32 (max_pos, 768);
 * h_out_dev fp32 (n_prompts*len, 768); xn_out_dev the same rows in `dtype` (the first projection's operand). */
int rz_text_embed_ln(int dtype, const int64_t* input_ids_dev, const float* word_emb_dev, const float* pos_emb_dev, const float* gamma_dev,
                     const float* beta_dev, float eps, float* h_out_dev, void* xn_out_dev, int n_prompts, int len, int vocab_size,
                     int max_position_embeddings, int pad_token_id, void* stream);
/* MPNet self-attention (TF:mpnet/modeling_mpnet.py:115-200): qkv_dev `dtype` (n_prompts*len, 3*heads*64) = q | k | v with q already
 * scaled by 1/sqrt(64); scores = q k^T + rel_bias[h][i][j] + (mask[t][j] ? 0 : -FLT_MAX); ctx_out_dev `dtype` (n_prompts*len, heads*64). */
int rz_text_attention(int dtype, const void* qkv_dev, const float* rel_bias_dev, const int64_t* attention_mask_dev, void* ctx_out_dev,
                      int n_prompts, int len, int heads, void* stream);
/* masked mean pool (modeling.py:148-156): out[t] = sum_i h[t][i] mask[t][i] / max(sum_i mask[t][i], 1e-9); h_dev fp32 (n_prompts, len, dim) */
int rz_masked_meanpool(const float* h_dev, const int64_t* attention_mask_dev, float* out_dev, int n_prompts, int len, int dim, void* stream);
/* The alignment heads beside VL-CABS — compute_logits_type "cls_alignment" / "global_alignment" (modeling.py:330-353: `image_cls_token @
 * key_features.T`, `image_features @ key_features.T`, einsum("ind,jd->ijn", image_patch_tokens, key_features[:, hidden:])) and the text
 * projector's nn.Linear (modeling.py:70-73, :199-200) — as one strided fp32 product, fixed summation order:
 *   out[(m / rows_per_group) * out_group_stride + (m % rows_per_group) * out_row_stride + n * out_col_stride] = sum_k a[m][k] b[n][k] (+ bias[n])
 * a_dev fp32 (M rows, leading dimension lda), b_dev fp32 (N rows, ldb), bias_dev fp32 (N) or NULL; K, lda, ldb multiples of 4. */
int rz_rows_dot(const float* a_dev, int64_t lda, const float* b_dev, int64_t ldb, const float* bias_dev, float* out_dev, int M, int N, int K,
                int rows_per_group, int64_t out_group_stride, int64_t out_row_stride, int64_t out_col_stride, void* stream);
/* image_features (modeling.py:113-117): F.normalize(cat([cls_token, patch_tokens.mean(dim=1)], dim=1), p=2, dim=1).  tokens_dev fp32, image b at
 * row b * image_stride_rows, row 0 = cls, rows 1 .. n_tokens-1 = patches, `dim` columns (multiple of 64); out_dev fp32 (batch, 2 * dim). */
int rz_image_features(const float* tokens_dev, int64_t image_stride_rows, int batch, int n_tokens, int dim, float* out_dev, void* stream);
/* Dinov2PatchEmbeddings + cls token + position embeddings (TF:dinov2/modeling_dinov2.py:97-149) as im2col + GEMM with the table epilogue:
 * pixel_values_dev fp32 (batch, channels, height, width); weight_dev `dtype` (768, k_pad) = the conv kernel flattened (c, ky, kx) and zero
 * padded to k_pad = round_up(channels*patch*patch, 64); table_dev fp32 (n_pad, 768) = row 0: cls + pos[0], rows 1..grid_h*grid_w: conv
 * bias + pos[t], 0 on pad rows; n_pad a multiple of 128 >= 1 + grid_h*grid_w; im2col_ws_dev: batch*n_pad*k_pad elements of `dtype`;
 * out_dev fp32 (batch*n_pad, 768) = the residual stream the first block reads. */
int rz_patch_embed(int dtype, const float* pixel_values_dev, int batch, int channels, int height, int width, int patch, const void* weight_dev,
                   int k_pad, const float* table_dev, int n_pad, void* im2col_ws_dev, float* out_dev, void* stream);

/* fp32 GEMM on the f16 / fp8 matrix pipes, kernel level — the two operand forms of the fp32 (1e-3) mode's vision encoder
 * (TF:dinov2/modeling_dinov2.py:199-213, :246-251, :281-297 computed with fp32 inputs): out_dev[m][n] += a[m][:] . w[n][:] + bias[n], fp32 `out_dev`
 * read-modify-write.  form 0 = every product as three f16 MFMAs over hi/lo planes; form 1 = "MX": a_hi b_hi on the f16 pipe + the two correction
 * terms as one block-scaled e4m3 MFMA.  a_dev (M, K), w_dev (N, K), bias_dev (N), ones_dev (N floats of 1.0), ws_a_dev / ws_w_dev: scratch of
 * 6 (form 0) or 4 (form 1) bytes per element of a / w.  M, N multiples of 256, K a multiple of 64, >= 128. */
int rz_gemm_f32_split(int form, const float* a_dev, const float* w_dev, const float* bias_dev, const float* ones_dev, float* out_dev,
                      void* ws_a_dev, void* ws_w_dev, int M, int N, int K, void* stream);

/* Tuning / A-B switches.  rz_set_option sets the PROCESS-WIDE value (what the measurement tools flip); rz_set_model_option sets one
 * handle's own value, which then overrides the process-wide one for that handle only (INT32_MIN = follow the process-wide value again):
 * two handles of one process can differ.  Defaults are the measured-fastest choices.
 *   "gemm_variant"     0 auto | 1 128x128 two-stage | 3 256x256 two-stage | 7 256x256 staggered 8-phase (16-bit, gemm7.hip)
 *                      | 8 the same K loop as a persistent kernel, one workgroup per CU (16-bit, gemm8.hip; default for big shapes).
 *                      10 / 11 / 12 (other K loops; two 256x128 workgroups per CU) are retired experiments: bit-identical, never faster
 *                      inside the step, compiled into the RZ_EXPERIMENTS tools library only — the product library runs 8 for them
 *   "gemm_v1_only"     (process-wide only) 1 = same as gemm_variant 1
 *   "gemm_small_tile"  the 128x128 kernel family (variant 1, every mode): 0 (default) = by grid size, n = tiles of 128x128 — up to 256 tiles (every
 *                      workgroup owns a CU) a four-stage LDS ring, three operand panel pairs in flight, on 128x128 tiles above 96 tiles, 128x64
 *                      (two waves) from 40, 64x64 (one wave) below; up to 340 tiles two stages on 128x64; above, two stages on 128x128 (rounds 1-5
 *                      everywhere).  1 / 2 / 3 force 128x128 / 64x64 / 128x64, + 20 / + 40 force two / four stages (tests, A/B).  Same K order in
 *                      each: bit-identical outputs
 *   "gemm_qkv_pair"    1 (default) = the block's q|k and v projections as ONE launch of the 128x128 kernel family, rastered as one GEMM of 3 D columns:
 *                      16-bit modes — where the persistent kernel's merged q|k|v projection does not apply (odd row counts) and, up to 448 tiles of
 *                      128x128 (one or two 518^2 images, up to eight 224^2 images), in front of it; fp32 mode (MX and three-plane forms, which have no
 *                      merged projection) — wherever each of the two would take that family alone.  2 = wherever the shapes allow (A/B);
 *                      0 = two launches (rounds 1-5).  Same tiles, same arithmetic: same bits
 *   "attn_variant"     0 default (16x16x32 MFMA, 4 waves x 32 query rows, row sums on the matrix pipe; bf16 without the running
 *                      maximum in the hot loop; round 6, bf16: 4 waves x 16 query rows — 64-row workgroups, same bits per row — where the grid
 *                      has fewer than 384 blocks of 128 rows, e.g. one 518^2 image) | 417 = the 128-row shape with the running maximum
 *                      tracked in every tile (what f16 always runs) | 401 / 402 = the 64- / 128-row workgroup forced (tests, A/B).
 *                      Other values run the default (the retired shapes live in the RZ_EXPERIMENTS tools library)
 *   "attn_f32_split"   1 (default) = fp32 mode runs attention as hi/lo-split f16 MFMAs; 0 = exact-fp32 MFMAs (16x16x4_f32)
 *   "gemm_f32_split"   1 (default) = fp32 mode runs the vision encoder's GEMMs as hi/lo-split f16 MFMAs; 0 = exact-fp32 MFMAs
 *   "gemm_f32_mx"      fp32 mode, hi/lo-split GEMMs: 1 (default) = wherever the shape allows (a launch's token rows a multiple of 256: every batch at 1024^2 and 518^2,
 *                      even batches at 224^2) the two correction terms
 *                      a_lo b_hi + a_hi b_lo run as ONE block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 planes with power-of-two
 *                      scales: fixed for activations, chosen per weight matrix from its largest |w|) beside the f16 a_hi b_hi MFMAs: 4 MFMA-units per 64 K instead of 6, 4 bytes per operand element
 *                      instead of 6, and the attention's second planes (q, k, V^T, P) likewise (32 f16 + 16 block-scaled MFMAs per wave and
 *                      64-key tile instead of 96); 2 = the same as 1 (rounds 4-5: "wherever the shape allows" against their default, which started the form at 64 row
 *                      tiles of 256); 3 = that former default (A/B); 0 = three f16 planes everywhere
 *   "attn_f32_mx"      fp32 mode, where "gemm_f32_mx" applies: 1 (default) = the attention's P V correction terms as block-scaled e4m3 MFMAs, the
 *                      scores Q K^T stay on three f16 planes (a score's error is exponentiated: 22 bits there); 2 = the scores' correction
 *                      terms too (2^-16 sum |q_d k_d| of error in a score: fine on small logits — the synthetic checkpoint's — only);
 *                      0 = three f16 planes throughout the attention
 *   "attn_f32_pv"      fp32 mode: 0 (default, "f32_precision high") = P V with its correction terms; 1 ("f32_precision fast") = P V as the single
 *                      product v_hi . p_hi (P and V at f16's 11 bits, row sums of the rounded P on the matrix pipe so that the weights still sum to
 *                      one): attention 89 -> 63 ms per B = 32 step at 1024^2 (198 -> 235 images/s), 3.1e-4 instead of 7e-5 from the reference
 *                      on the goldens but 1.2e-3 on the outlier-channel checkpoint G8 — outside the 1e-3 contract there, hence not the default
 *                      (profiles/r05/fp32_term_ablation.log).  With 1, "attn_f32_mx" only chooses the scores' form (0 / 1: f16 planes, 2: e4m3 pairs)
 *   "f32_drop"         fp32 mode, ACCURACY ABLATION only (tools/fp32_term_ablation.py; needs gemm_f32_mx = 0): bit mask of GEMM classes computed on
 *                      their hi planes alone — 1 q|k projection, 2 V projection, 4 out-projection, 8 fc1, 16 fc2, 32 patch embedding
 *   "f32_split_guard"  1 (default) = fp32 mode: a forward in which a value left the range of the hi/lo planes (|x| > 65504) is
 *                      repeated on the exact-fp32 kernels: rz_vision_forward enqueues that second pass behind the first with the
 *                      device guard word as every launch's predicate (its waves leave at once when the word is 0: ~75 empty
 *                      launches) — no host read, so the call stays asynchronous and a captured forward carries the guard into
 *                      every replay; rz_get_model_option(h, "f32_split_guard_reruns") reads the device-side count of repeats
 *                      (it waits for the device first: not while a stream is capturing)
 *   "ln_fused"         1 (default) = the blocks' LayerNorms are fused into the GEMMs either side of them (16-bit modes);
 *                      0 = stand-alone LayerNorm kernels everywhere
 *   "sim_op"           VL-CABS similarity (losses.py:207-217): 0 (default) "cos" — the released config; 1 "dot" — RadZeroLoss's constructor
 *                      default: LayerNorm without L2 normalisation, scores / sqrt(hidden), both sides normalised in the final logit.
 *                      (A checkpoint tensor "loss_fns.RadZeroLoss.attn_temperature", when loaded, is the score temperature of "cos".)
 *   "pad_rows"         token rows per image are padded to: 0 (default) a multiple of 128, of 256 where that costs < 2 % more rows
 *                      | 128 | 256 always that multiple.  Setting it on a handle drops its position tables and workspace sizes
 *                      (call rz_set_position_table / rz_reserve again).
 * (The batch-chunking / two-stream / tile-walk switches of rounds 1-4 — vision_chunk, vision_streams, mlp_chunk, gemm_raster — measured
 * no gain and are known to the RZ_EXPERIMENTS tools library only.) */
int rz_set_option(const char* name, int value);
int rz_set_model_option(rz_handle_t h, const char* name, int value);
/* the value in force for this handle (its own, else the process-wide one); also "f32_split_guard_reruns" and the read-only facts of the last
 * rz_vision_forward: "last_batch", "last_npad" (token rows per image after padding), "last_f32_form" (0 the dtype's own kernels | 1 three f16 planes | 2 MX form) */
int rz_get_model_option(rz_handle_t h, const char* name, int* value_out);
/* diagnostic library builds only (-DRZ_EXPERIMENTS, tools/kstamp8.py): "gemm_v8_stamps" = device buffer of 256 x 8 x 32 uint64 that the
 * stamped build of the persistent GEMM fills with per-wave K-loop / epilogue times; the production library knows no buffer */
int rz_debug_buffer(const char* what, void* dev_ptr);

/* ---- measurement: HIP-event timing of kernel families on the launch stream ---- */
enum rz_prof_family { RZ_PROF_ATTN = 0, RZ_PROF_GEMM = 1, RZ_PROF_ROWOPS = 2, RZ_PROF_VLCABS = 3, RZ_PROF_POST = 4 /* map upsample / grounding */, RZ_PROF_NFAM = 5 };
/* enable: 0 off | 1 every family | 1 + (mask << 1): only the families whose bit (1 << rz_prof_family) is set in mask */
int rz_profile_enable(rz_handle_t h, int enable);
/* after a stream synchronize: total milliseconds and launch count per family (arrays of RZ_PROF_NFAM) since enable; resets them */
int rz_profile_read(rz_handle_t h, float* ms_per_family, int64_t* launches_per_family);

#ifdef __cplusplus
}
#endif
#endif /* RADZERO_HIP_H */
