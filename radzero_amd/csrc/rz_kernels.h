// radzero_hip — internal kernel launcher interface (host side).  The public C-ABI is include/radzero_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rz {

enum Epilogue : int {
    EPI_STORE = 0,        // out<T>[m][n] = acc + bias
    EPI_GELU = 1,         // out<T>[m][n] = gelu_erf(acc + bias)
    EPI_HEADS = 2,        // out<T>[b][head][tok][64] = acc + bias            (q | k projection)
    EPI_VT = 3,           // out<T>[b][head][d][tok]  = acc + bias            (v projection, transposed)
    EPI_RESID_SCALE = 4,  // resid[m][n] += scale[n] * (acc + bias)          (LayerScale + residual, pre-LN blocks)
    EPI_RESID_ADD = 5,    // out_f32[m][n] = acc + bias + resid[m][n]        (post-LN blocks)
    EPI_PATCH = 6,        // out_f32[m][n] = acc + scale[tok][n]             (patch-embed: pos/cls/bias table)
    EPI_STORE_F32 = 7,    // out_f32[m][n] = acc + bias
    EPI_QKV = 8,          // merged q|k|v projection: columns < split_n as EPI_HEADS into `out`, the rest as EPI_VT into `out2` (gemm8.hip)
    // ---- LayerNorm fused into the GEMMs either side of it (gemm8.hip only; see "Fused LayerNorm" there) ----
    EPI_RESID_SCALE_LN = 9,   // EPI_RESID_SCALE + the next LayerNorm's inputs: ln_hb<T>[m][n] = ln_gamma[n] * new resid, ln_part[m][12] = (mean, M2) of 64-column slices
    EPI_QKV_LN = 10,          // EPI_QKV on the UN-normalised operand: (acc - mu[m]*scale[n]) * rstd[m] + bias[n], (mu, rstd) = ln_stat[m]
    EPI_GELU_LN = 11,         // EPI_GELU likewise
    EPI_HEADS_LN = 12,        // EPI_HEADS / EPI_VT likewise: the two halves of EPI_QKV_LN as separate launches of the 128x128 kernel
    EPI_VT_LN = 13,           //   (small batches, where the persistent 256x256 kernel would leave CUs idle)
    EPI_PATCH_LN = 14,        // EPI_PATCH + the first LayerNorm's inputs as EPI_RESID_SCALE_LN writes them (centring constant 0; out = fp32 [M][N], ln_mu unused): gemm8.hip only
};

struct GemmArgs {
    const void* A; int64_t lda;   // [M][K], leading dim in elements
    const void* W; int64_t ldw;   // [N][K]
    int M, N, K;
    const float* bias;            // [N] or nullptr
    void* out; int64_t ldo;
    const float* scale;           // EPI_RESID_SCALE: lambda[N]; EPI_PATCH: table [rows_per_image][N]
    float* resid; int64_t ldr;    // fp32 residual stream
    int rows_per_image;           // padded tokens per image (EPI_HEADS / EPI_VT / EPI_PATCH)
    int heads_total;              // heads in the destination tensor (EPI_HEADS / EPI_VT)
    void* out2 = nullptr;         // EPI_QKV: transposed-v destination
    int heads_total2 = 0;         // EPI_QKV: heads in `out2`
    int split_n = 0;              // EPI_QKV: first column of the v block (multiple of 256)
    const float* ln_stat = nullptr;  // EPI_*_LN consumers: [M][2] = (mean - centring constant, rstd) of every operand row; `scale` = c1[N], `bias` = c2[N]
    float* ln_part = nullptr;        // EPI_RESID_SCALE_LN: [M][12][2] partial statistics (N must be 768)
    void* ln_hb = nullptr;           // EPI_RESID_SCALE_LN: [M][N] copy of the new residual TIMES ln_gamma[n], in the compute dtype
    const float* ln_gamma = nullptr; // EPI_RESID_SCALE_LN: gain of the LayerNorm that will consume ln_hb
    const float* ln_mu = nullptr;    // EPI_RESID_SCALE_LN: [M] centring constant of each row (its mean before this update): ln_hb = T((x - ln_mu[m]) * ln_gamma[n])
    int64_t plane_off = 0;        // hi/lo-split outputs (fp32 mode, EPI_HEADS / EPI_VT): elements from the hi plane to the lo plane
    unsigned* ovf_flag = nullptr; // hi/lo-split outputs: word that receives 1 when a value leaves the f16 range (rz_common.h flag_f16_range)
    int mx_w_e8_hi = 123, mx_w_e8_lo = 112;   // fp32 mode, MX form: E8M0 scale bytes (127 + log2 scale) of THIS weight matrix's hi8 / lo8 planes; defaults = 2^-4 / 2^-15 (rz_common.h)
    const unsigned* run_if = nullptr;  // exact-fp32 kernels only (fp32 mode's overflow guard, api.hip rz_vision_forward): non-null => the launch does nothing unless *run_if != 0
    int small_tile = 0;           // 128x128-kernel family (gemm.hip): 0 by grid size | 1 128 x 128 (4 waves) | 2 64 x 64 (1 wave) | 3 128 x 64 (2 waves); + 10 S forces a ring of S = 2 / 4 panel pairs
    int raster = 0;               // gemm12.hip tile order inside an XCD: 0 = gemm8's (4 x tiles_n groups) | S > 0 = slab walk, slabs of <= S n tiles
    int variant = 0;              // kernel choice: 0 auto | 1 128x128 two-stage | 3 256x256 two-stage | 7 staggered 8-phase (gemm7.hip) | 8 persistent (gemm8.hip) | 10 persistent, 4 waves x 128x128, asm K loop (gemm10.hip) | 11 persistent, 8 waves, one phase per K tile (gemm11.hip)
};

hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t s);
#ifdef RZ_EXPERIMENTS
void gemm_v8_set_stamp_buffer(void* dev_u64);   // diagnostic build: non-null => the stamped persistent kernel writes 256 x 8 x 32 u64 there
#endif
bool gemm_v7_ok(int dtype, const GemmArgs& g);
bool gemm_v8_ok(int dtype, int epi, const GemmArgs& g);
bool gemm_v8_mx_ok(int epi, int out_kind, const GemmArgs& g);          // fp32 mode's MX form on the persistent kernel (gemm8.hip): which epilogues / output forms / shapes
hipError_t launch_gemm_v8_mx(int epi, int out_kind, const GemmArgs& g, hipStream_t s);
bool gemm_qkv_fused_ok(int dtype, const GemmArgs& g);
// two GEMMs over the same rows (A, M, K) in one launch of the 128x128 family (gemm.hip gemm_pair_kernel): (EPI_HEADS_LN, EPI_VT_LN) or (EPI_HEADS, EPI_VT), 16-bit modes,
// where each of the two would have gone to that family on its own; same bits as the two launches
bool gemm_pair_ok(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb);
hipError_t launch_gemm_pair(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb, hipStream_t s);
// fp32 mode's split forms of the same pair (EPI_HEADS -> hi / lo f16 planes, EPI_VT -> v_kind 1 hi / lo f16 planes | 3 hi f16 + e4m3 pair plane): form 0 = three f16 planes
// along K (operands as for launch_gemm_split_f32out), 1 = MX (as for launch_gemm_small_mx)
bool gemm_pair_f32_ok(int form, const GemmArgs& ga, const GemmArgs& gb, int v_kind);
hipError_t launch_gemm_pair_f32(int form, const GemmArgs& ga, const GemmArgs& gb, int v_kind, hipStream_t s);
bool gemm_patch_ln_ok(int dtype, const GemmArgs& g);                  // may the patch-embedding GEMM write block 0's LayerNorm inputs itself (EPI_PATCH_LN)
bool gemm_ln_fused_ok(int dtype, int M, int D, int F, int variant);   // may a Dinov2 block of M token rows use the fused-LayerNorm epilogues
// fp32 mode on the f16 matrix pipe: operands split into f16 planes along K (gemm.hip)
hipError_t launch_gemm_split_f32out(int epi, const GemmArgs& g, hipStream_t s, bool split_out = false);
hipError_t launch_split3(const float* src, int64_t ld, void* dst, int64_t rows, int K, int w_layout, unsigned* ovf_flag, hipStream_t s, int w_e8_hi = 123);   // w_layout 0 / 1: f16 planes [hi|lo|hi] / [hi|hi|lo]; 2 / 3: MX form of an activation / weight matrix (K % 64 == 0); w_e8_hi: layout 3's hi8 plane scale as an E8M0 byte (lo8: 2^-11 of it)
hipError_t launch_absmax_bits(const float* src, int64_t n, unsigned* out_bits, hipStream_t s);   // *out_bits = max over src of the bit pattern of |x| (NaN / inf included: the largest patterns); out_bits zeroed by the caller
hipError_t launch_gemm_v8(int dtype, int epi, const GemmArgs& g, hipStream_t s);   // persistent 256x256 kernel (gemm8.hip)
bool gemm_v10_ok(int dtype, int epi, const GemmArgs& g);
bool gemm_v11_ok(int dtype, int epi, const GemmArgs& g);
hipError_t launch_gemm_v11(int dtype, int epi, const GemmArgs& g, hipStream_t s);  // persistent, 8 waves, ONE LOAD / MFMA phase per K tile (gemm11.hip)
hipError_t launch_gemm_v10(int dtype, int epi, const GemmArgs& g, hipStream_t s);  // persistent, 4 waves x 128x128, asm K loop (gemm10.hip)
bool gemm_v12_ok(int dtype, int epi, const GemmArgs& g);
hipError_t launch_gemm_v12(int dtype, int epi, const GemmArgs& g, hipStream_t s);  // persistent, two 256x128 workgroups per CU (gemm12.hip)
hipError_t launch_gemm_v7(int variant, int dtype, int epi, const GemmArgs& g, hipStream_t s);
hipError_t launch_gemm_v7_f16_out(int epi, const GemmArgs& g, bool split_out, hipStream_t s);   // f16 operands, fp32 / hi-lo-split outputs
// fp32 mode, MX form (rz_common.h "MX form"): f16 hi plane + block-scaled fp8 correction planes; out_kind 0 fp32 RMW / table, 1 hi/lo planes, 2 MX A operand
bool gemm_v7_mx_ok(const GemmArgs& g);
bool gemm_small_mx_ok(int epi, int out_kind, const GemmArgs& g);      // the same form on the 128 x 128 kernel (gemm.hip; round 6): small shapes
bool gemm_small_mx_pays(int epi, const GemmArgs& g);
hipError_t launch_gemm_small_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t s);
hipError_t launch_gemm_v7_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t s);

// Flash attention over per-head tensors: q,k [B][H][Npad][64], vT [B][H][64][Npad] -> ctx [B*Npad][H*64].
// Scores are NOT rescaled inside (1/sqrt(dh) is folded into the packed q weights).
// q/k of image b start at q + b*qk_batch_stride (elements), heads contiguous ([H][Npad][64]).
size_t flash_attn_split_workspace_bytes(int B, int H, int n_pad);
hipError_t launch_flash_attn_f32_split(const float* q, const float* k, const float* vT, float* ctx, void* split_ws, int64_t qk_batch_stride,
                                       int B, int H, int n_valid, int n_pad, unsigned* ovf_flag, hipStream_t s, int mxa = 0, int pv_hi = 0);   // mxa: correction terms as block-scaled e4m3 MFMAs
hipError_t launch_flash_attn_split_planes(const void* q_hi, const void* k_hi, const void* v_hi, void* ctx3, int64_t qk_batch_stride,
                                          int64_t qk_lo_off, int64_t v_lo_off, int B, int H, int n_valid, int n_pad, unsigned* ovf_flag, hipStream_t s,
                                          int mx_out = 0, int mxa = 0, int abl = 0);      // mx_out: ctx in the MX form (4 bytes per element) instead of [hi | lo | hi] f16;
                                                                             // mxa: the second planes are e4m3 pair planes (attention.hip "MXA")
hipError_t launch_flash_attn(int dtype, const void* q, const void* k, const void* vT, void* ctx,
                             int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, int waves, hipStream_t s, const unsigned* run_if = nullptr);   // run_if: fp32 operands only, see GemmArgs::run_if

// MPNet self-attention for short sequences: qkv [T*L][3*H*64] (q|k|v), additive relative-position bias expanded by the
// host to rel_bias[H][L][L], key-padding mask [T][L]; ctx [T*L][H*64].
hipError_t launch_text_attn(int dtype, const void* qkv, const float* rel_bias, const int64_t* attn_mask, void* ctx, int T, int L,
                            int H, hipStream_t s, const unsigned* run_if = nullptr, void* planes_out = nullptr, unsigned* ovf_flag = nullptr);      // planes_out (fp32 operands only): ctx leaves as [rows][3 H 64] f16 = [hi | lo | hi] instead (the text encoder's out-projection operand in the fp32 mode)

// Fused-LayerNorm support (rowops.hip).  The T copy of a residual row is CENTRED before rounding: (x - c_m) * gain with c_m a
// per-row constant close to the row mean (its mean before the last residual update, kept in mu[rows]); consumers get
// stat [rows][2] = (mean - c_m, rstd).  ln_finalize: partial statistics [rows][12][2] (mean, M2 of 64-column slices, as
// EPI_RESID_SCALE_LN writes them) + the constant that producer used (mu_inout) -> stat, and mu_inout = the new mean.  ln_prepare: rows of 768 fp32 -> T copy + stat; with
// gamma/beta != nullptr the row is first LayerNorm'ed in place (out_f32, may alias in) and copy / stat describe the result.
// The T copy is multiplied by `copy_gain` (the gain of the LayerNorm that will consume it).
hipError_t launch_ln_finalize(const float* part, float* mu_inout, float* stat, float eps, int64_t rows, hipStream_t s, bool mu_is_zero = false);
hipError_t launch_ln_prepare(int dtype, const float* in, const float* gamma, const float* beta, float eps_in, float* out_f32,
                             const float* copy_gain, void* copy_t, float* mu_out, float* stat, float eps_stat, int64_t rows, int D, hipStream_t s);

// Alignment heads beside VL-CABS (modeling.py:330-353, :115-117), fp32: out[(m / rpg) * og + (m % rpg) * orow + n * ocol] = a[m] . b[n] (+ bias[n]);
// image_features = l2norm([cls | mean of the patch tokens]) -> out [B][2 D]
hipError_t launch_rows_dot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* out, int M, int N, int K,
                           int rows_per_group, int64_t out_group_stride, int64_t out_row_stride, int64_t out_col_stride, hipStream_t s);
hipError_t launch_image_features(const float* tokens, int64_t image_stride, int B, int n_tokens, int D, float* out, hipStream_t s);

// LayerNorm over rows of 768: fp32 in; writes T-typed normalized copy (out_t, may be null) and/or
// fp32 (out_f32, may alias in).
hipError_t launch_layernorm_split3(const float* in, const float* gamma, const float* beta, float eps, void* out3, int64_t rows, int D, unsigned* ovf_flag, hipStream_t s, int mx = 0, float* out_f32 = nullptr);   // mx: the MX form (rz_common.h), 4 D bytes per row
hipError_t launch_layernorm(int dtype, const float* in, const float* gamma, const float* beta, float eps,
                            void* out_t, float* out_f32, int64_t rows, int D, hipStream_t s, const unsigned* run_if = nullptr);

// fp32 mode's overflow guard (api.hip rz_vision_forward): words[0] = the flag the plane producers raise, words[4] = forwards repeated so far.
// op 0: words[flag_idx] = 0 (start of a forward); op 1: if (words[flag_idx]) ++words[count_idx] (behind the predicated exact-fp32 pass)
hipError_t launch_guard_word(unsigned* words, int op, hipStream_t s, int flag_idx = 0, int count_idx = 4);      // (5, 6): the text encoder's flag / counter

// im2col for the 14x14/stride-14 patch conv: pixels fp32 [B][C][H][W] -> A<T>[B][Npad][Kpad];
// row 0 (CLS) and rows >= 1+gh*gw are zero; columns >= C*14*14 are zero.
hipError_t launch_im2col(int dtype, const float* px, void* out, int B, int C, int Himg, int Wimg, int patch,
                         int gh, int gw, int n_pad, int k_pad, hipStream_t s, const unsigned* run_if = nullptr);

// MPNet embeddings: word_emb[ids] + pos_emb[pos_ids(ids)] -> LN -> h fp32 + T copy.
hipError_t launch_text_embed(int dtype, const int64_t* ids, const float* word_emb, const float* pos_emb,
                             const float* gamma, const float* beta, float eps, float* h, void* xn,
                             int T, int L, int D, int vocab, int max_pos, int pad_id, hipStream_t s, const unsigned* run_if = nullptr);

// masked mean pool: h [T][L][D] fp32, mask [T][L] -> out [T][D]
hipError_t launch_masked_meanpool(const float* h, const int64_t* mask, float* out, int T, int L, int D, hipStream_t s);

// shared LN + L2 normalise rows (fp32 -> fp32).  in rows have stride ld_in.
hipError_t launch_ln_l2norm(const float* in, int64_t ld_in, const float* gamma, const float* beta, float eps,
                            float* out, int64_t rows, int D, int l2, hipStream_t s);

// VL-CABS: tokens [B][Npad][D] fp32 (shared LN + L2 normalisation applied inside), qhat [T][D] -> scores [B][T][N] (= cos/tau),
// t2i_logits [T][B], logits [B][T] (= t2i^T / tau).  ws: float workspace.
size_t vlcabs_workspace_floats(int B, int T, int n_pad, int D);
// sim_dot = 0: cosine (tokens and queries L2-normalised), scores / score_denominator (= tau); 1: losses.py's "dot": LayerNorm only,
// scores / score_denominator (= sqrt(D)), both sides normalised in the final logit.  logits = t2i^T / logit_tau.
hipError_t launch_vlcabs(const float* tokens, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* qhat, float score_denominator, float logit_tau, int sim_dot, float* ws, float* scores, float* t2i_logits,
                         float* logits, int B, int T, int n_valid, int n_pad, int D, hipStream_t s);

// bilinear upsample (align_corners=False) of patch-grid maps [M][g][g] -> [M][H][W], optional sigmoid;
// optional per-map argmax (flat index of first max) -> argmax_out[M].
hipError_t launch_upsample_bilinear(const float* maps, int64_t map_stride, float* out, int64_t* argmax_out, int M, int g,
                                    int Hout, int Wout, int apply_sigmoid, int keep_aspect, hipStream_t s);

// argmax (x, y) of the bilinear-upsampled map, without materialising it; keys_ws: M x u64 scratch
hipError_t launch_grounding_points(const float* maps, int64_t map_stride, unsigned long long* keys_ws, int* xy_out, int M, int g,
                                   int Hout, int Wout, int keep_aspect, hipStream_t s);

// device-side image preprocessing (preprocess.hip): raw image [H][W][C] (src_dtype 0 u8 / 1 u16 / 2 f32) -> fp32 [3][S][S].
// bounds_*/kk_*: Pillow resampling tables (device, int32): bounds [out][2] = (first, count), kk [out][ksize] 22-bit fixed point.
// ws8: scratch of H*W*C + H*S*C + S*S*C bytes; mm: 2 x u32 scratch.
hipError_t launch_preprocess(const void* img, int src_dtype, int H, int W, int C, int S, const int* bounds_h, const int* kk_h, int ksize_h,
                             const int* bounds_v, const int* kk_v, int ksize_v, const float* mean, const float* stdv, float rescale,
                             unsigned char* ws8, unsigned* mm, float* out, int minmax_normalize, hipStream_t s);

// batched form: `descs_host` = n internal descriptors (preprocess.hip PreDesc, built by api.hip), copied to the head of `ws`
size_t preprocess_batch_desc_bytes(int n);
hipError_t launch_preprocess_batch(const void* descs_host, int n, int max_ph, int S, const float* mean, const float* stdv, float rescale,
                                   unsigned char* ws, float* out, int minmax_normalize, hipStream_t s);

// strided gather of valid tokens: src [B][Npad][D] -> dst [B][N][D]
hipError_t launch_copy_tokens(const float* src, float* dst, int B, int n_valid, int n_pad, int D, hipStream_t s, const unsigned* run_if = nullptr);

}  // namespace rz
