// Host-side sanitizer run (tools/host_asan.sh): the C-ABI's host code under AddressSanitizer + UBSan + LeakSanitizer, for all three
// compute dtypes: create -> load the synthetic checkpoint -> weights_ready -> PARTIAL reload (a gain, one q matrix, fc1, the text side,
// temperatures) -> weights_ready -> options -> position tables for two grids -> reserve (grow twice) -> vision / text / VL-CABS /
// upsample / grounding / batched preprocessing with the range-checking launcher stubs -> error paths -> destroy.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/radzero_hip.h"

namespace rz { int host_asan_ranges_checked(); }

struct Tensor { std::string name; std::vector<float> data; };
static std::vector<Tensor> read_weights(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) exit(2);
    std::vector<Tensor> out(n);
    for (auto& t : out) {
        uint32_t len; uint64_t numel;
        if (fread(&len, 4, 1, f) != 1) exit(2);
        t.name.resize(len);
        if (fread(&t.name[0], 1, len, f) != len || fread(&numel, 8, 1, f) != 1) exit(2);
        t.data.resize(numel);
        if (fread(t.data.data(), 4, numel, f) != numel) exit(2);
    }
    fclose(f);
    return out;
}
#define OK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rz_last_error()); exit(3); } } while (0)
#define FAILS(call) do { int rc_ = (call); if (rc_ == 0) { fprintf(stderr, "%s unexpectedly succeeded\n", #call); exit(4); } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: driver weights.bin\n"); return 2; }
    const auto sd = read_weights(argv[1]);
    for (int dt = 0; dt < 3; ++dt) {
        rz_config cfg = {dt, 768, 12, 4, 14, 3, 2, 1, 1e-6f, 512, 514, 1, 3072, 1e-5f, 1, 1e-5f};
        rz_handle_t h = nullptr;
        OK(rz_create(&cfg, &h));
        FAILS(rz_weights_ready(h));                                           // nothing loaded yet
        for (const auto& t : sd) OK(rz_load_weight(h, t.name.c_str(), t.data.data(), (int64_t)t.data.size()));
        OK(rz_weights_ready(h));
        // partial reload: the tensors that invalidate the fused-LayerNorm fold, the split-weight copies and the temperatures
        for (const auto& t : sd)
            if (t.name.find("layer.1.norm1.weight") != std::string::npos || t.name.find("layer.0.attention.attention.query.weight") != std::string::npos ||
                t.name.find("layer.0.mlp.fc1.weight") != std::string::npos || t.name.find("text_model.encoder.layer.0.attention.attn.q.weight") != std::string::npos ||
                t.name.find("loss_temperature") != std::string::npos || t.name.find("patch_embeddings.projection.weight") != std::string::npos)
                OK(rz_load_weight(h, t.name.c_str(), t.data.data(), (int64_t)t.data.size()));
        FAILS(rz_load_weight(h, "vision_model.encoder.layer.0.norm1.weight", sd[0].data.data(), 3));      // wrong numel
        FAILS(rz_load_weight(h, "vision_model.encoder.layer.7.norm1.weight", sd[0].data.data(), 768));    // no such layer
        OK(rz_weights_ready(h));
        OK(rz_set_model_option(h, "ln_fused", dt == 2 ? 0 : 1));
        OK(rz_set_model_option(h, "pad_rows", 256));
        OK(rz_set_model_option(h, "pad_rows", INT32_MIN));
        FAILS(rz_set_model_option(h, "no_such_option", 1));
        int v = -1;
        OK(rz_get_model_option(h, "gemm_variant", &v));
        for (int grid : {16, 19}) {                                           // 224^2 and 266^2
            std::vector<float> pos((size_t)(1 + grid * grid) * 768, 0.01f);
            OK(rz_set_position_table(h, grid, grid, pos.data()));
        }
        OK(rz_reserve(h, 2, 257, 3, 9));
        OK(rz_reserve(h, 3, 362, 5, 12));                                     // grows
        const int B = 3, S = 266, N = 362, T = 5, L = 12;
        std::vector<float> px((size_t)B * 3 * S * S, 0.5f), tokens((size_t)B * N * 768), feats((size_t)T * 768), rel((size_t)12 * L * L, 0.f);
        std::vector<int64_t> ids((size_t)T * L, 5), mask((size_t)T * L, 1);
        OK(rz_vision_forward(h, px.data(), B, 3, S, S, tokens.data(), nullptr));
        OK(rz_vision_forward(h, px.data(), 2, 3, 224, 224, nullptr, nullptr));
        FAILS(rz_vision_forward(h, px.data(), 2, 4, 224, 224, nullptr, nullptr));      // wrong channel count
        FAILS(rz_vision_forward(h, px.data(), 9, 3, 224, 224, nullptr, nullptr));      // beyond the reserved batch
        OK(rz_set_model_option(h, "gemm_f32_mx", 2));                          // fp32 mode: the MX form wherever the rows are a multiple of 256 (2 x 384)
        OK(rz_vision_forward(h, px.data(), 2, 3, 224, 224, nullptr, nullptr));
        OK(rz_set_model_option(h, "gemm_f32_mx", 0));
        OK(rz_vision_forward(h, px.data(), 2, 3, 224, 224, nullptr, nullptr));
        OK(rz_set_model_option(h, "gemm_f32_mx", INT32_MIN));
        OK(rz_vision_forward(h, px.data(), B, 3, S, S, nullptr, nullptr));
        OK(rz_text_forward(h, ids.data(), mask.data(), T, L, rel.data(), feats.data(), nullptr));
        std::vector<float> scores((size_t)B * T * N), t2i((size_t)T * B), logits((size_t)B * T);
        OK(rz_vlcabs(h, feats.data(), T, B, scores.data(), t2i.data(), logits.data(), nullptr));
        std::vector<float> maps((size_t)B * T * 300 * 280);
        OK(rz_upsample_maps_ex(h, scores.data() + 1, N, B * T, 19, 300, 280, 1, 1, maps.data(), nullptr));
        std::vector<int32_t> xy((size_t)B * T * 2);
        std::vector<uint64_t> keys((size_t)B * T);
        OK(rz_grounding_points_ex(h, scores.data() + 1, N, B * T, 19, 300, 280, 0, xy.data(), keys.data(), nullptr));
        // batched preprocessing: 20 ragged images (> 16: two descriptor uploads), aspect-ratio padding on the odd ones
        const int NI = 20, OUT = 224;
        std::vector<std::vector<uint16_t>> raws(NI);
        std::vector<std::vector<int32_t>> bnd(NI * 2), kk(NI * 2);
        std::vector<rz_image_desc> descs(NI);
        for (int i = 0; i < NI; ++i) {
            const int hgt = 200 + 7 * i, wid = 260 - 5 * i, side = i & 1 ? (hgt > wid ? hgt : wid) : 0;
            raws[i].assign((size_t)hgt * wid, 100);
            const int ph = side ? side : hgt, pw = side ? side : wid, ksh = 2 * ((2 * pw + OUT - 1) / OUT) + 1, ksv = 2 * ((2 * ph + OUT - 1) / OUT) + 1;
            bnd[2 * i].assign((size_t)OUT * 2, 0); kk[2 * i].assign((size_t)OUT * ksh, 0);
            bnd[2 * i + 1].assign((size_t)OUT * 2, 0); kk[2 * i + 1].assign((size_t)OUT * ksv, 0);
            for (int o = 0; o < OUT; ++o) { bnd[2 * i][2 * o + 1] = 1; bnd[2 * i + 1][2 * o + 1] = 1; }
            rz_image_desc d = {};
            d.image_dev = raws[i].data(); d.src_dtype = 1; d.height = hgt; d.width = wid; d.channels = 1;
            d.pad_left = side ? (side - wid) / 2 : 0; d.pad_top = side ? (side - hgt) / 2 : 0; d.padded_height = ph; d.padded_width = pw;
            d.bounds_h_dev = bnd[2 * i].data(); d.coeffs_h_dev = kk[2 * i].data(); d.ksize_h = ksh;
            d.bounds_v_dev = bnd[2 * i + 1].data(); d.coeffs_v_dev = kk[2 * i + 1].data(); d.ksize_v = ksv;
            descs[i] = d;
        }
        const size_t need = rz_preprocess_batch_workspace(descs.data(), NI, OUT);
        if (!need) { fprintf(stderr, "rz_preprocess_batch_workspace refused the batch: %s\n", rz_last_error()); return 5; }
        std::vector<unsigned char> ws(need);
        std::vector<float> pv((size_t)NI * 3 * OUT * OUT);
        const float mean[3] = {0.48f, 0.45f, 0.41f}, sdv[3] = {0.27f, 0.26f, 0.28f};
        OK(rz_preprocess_batch(descs.data(), NI, OUT, mean, sdv, 1.f / 255.f, 1, ws.data(), ws.size(), pv.data(), nullptr));
        FAILS(rz_preprocess_batch(descs.data(), NI, OUT, mean, sdv, 1.f / 255.f, 1, ws.data(), ws.size() / 2, pv.data(), nullptr));
        OK(rz_profile_enable(h, 1));
        OK(rz_vision_forward(h, px.data(), 1, 3, 224, 224, nullptr, nullptr));
        float ms[8]; int64_t launches[8];
        OK(rz_profile_read(h, ms, launches));
        OK(rz_destroy(h));
        printf("[host_asan] dtype %d: clean, %d kernel ranges checked so far\n", dt, rz::host_asan_ranges_checked());
    }
    printf("[host_asan] OK\n");
    return 0;
}
