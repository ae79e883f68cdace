"""CPU ORACLE for RadZero's VL-CABS inference path  —  TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
The product path (`radzero_amd/`) never imports it and fails loudly if the HIP library is missing.

What it is: a from-the-equations restatement (PyTorch CPU fp32, plain tensor ops) of the arithmetic
the reference executes on this path.  Most of that arithmetic lives in a third-party dependency that
is NOT vendored in /root/reference: HuggingFace `transformers` (pinned ==4.39.3, requirements.txt:247;
5.15.0 installed here), classes Dinov2Model / Dinov2Encoder / MPNetModel.  `TF:` below cites the
installed transformers source (`transformers/models/...`), other citations are relative to
/root/reference.

Pinning: the reference ships no tests / golden vectors for this path, so the oracle
is pinned against outputs of the reference itself, run in the build container by
tools/make_goldens.py (reference imported from where it lies, synthetic checkpoint from
radzero_amd/weights.py) and committed as tests/golden/*.npz.  tests/test_oracle_golden.py checks
every fixture (max-abs <= 2e-4 on cosine/0.07-scaled scores; typically ~1e-5).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim, biased variance."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu_erf(x):
    """hidden_act 'gelu' = exact erf form (TF:activations GELUActivation)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def l2_normalize(x, eps=1e-12):
    """F.normalize(p=2, dim=-1): x / max(||x||, eps)."""
    n = torch.sqrt((x * x).sum(-1, keepdim=True))
    return x / torch.clamp(n, min=eps)


def attention(q, k, v, scale, bias=None, impl="eager"):
    """softmax(q k^T * scale + bias) v over (..., heads, L, dh).
    TF:dinov2/modeling_dinov2.py:153-178 (eager_attention_forward); TF:mpnet/modeling_mpnet.py:150-171."""
    if impl == "sdpa":
        return F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=scale)
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v)


# --------------------------------------------------------------------------------------------
# vision: Dinov2 embeddings + 12 pre-LN blocks + final LN + 2 align blocks
# --------------------------------------------------------------------------------------------
def interpolate_pos_encoding(position_embeddings, grid_h, grid_w):
    """TF:dinov2/modeling_dinov2.py:57-95.  position_embeddings (1, 1+g0*g0, D) -> (1, 1+gh*gw, D).
    Bicubic, align_corners=False, fp32; CLS position kept.  Returned unchanged when the grid matches."""
    pe = _t(position_embeddings)
    npos = pe.shape[1] - 1
    g0 = int(round(npos ** 0.5))
    if grid_h * grid_w == npos and grid_h == grid_w:
        return pe
    d = pe.shape[-1]
    cls_pos = pe[:, :1]
    patch_pos = pe[:, 1:].reshape(1, g0, g0, d).permute(0, 3, 1, 2)
    patch_pos = F.interpolate(patch_pos.float(), size=(grid_h, grid_w), mode="bicubic", align_corners=False)
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, d)
    return torch.cat([cls_pos, patch_pos], dim=1)


def patch_embed(pixel_values, w, b, cls_token, pos):
    """TF:dinov2/modeling_dinov2.py:98-149: Conv2d(k=14,s=14) as a GEMM over non-overlapping patches
    (trailing pixels beyond g*14 are dropped by the strided conv), prepend CLS, add pos-embed."""
    bsz, c, hh, ww = pixel_values.shape
    p = w.shape[-1]
    if c != w.shape[1]:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
    gh, gw = hh // p, ww // p
    x = pixel_values[:, :, : gh * p, : gw * p].reshape(bsz, c, gh, p, gw, p)
    x = x.permute(0, 2, 4, 1, 3, 5).reshape(bsz, gh * gw, c * p * p)      # (B, Np, 588) in (c, ky, kx) order
    emb = x @ w.reshape(w.shape[0], -1).t() + b
    emb = torch.cat([cls_token.expand(bsz, -1, -1), emb], dim=1)
    return emb + pos


def dino_block(h, P, prefix, n_heads, eps, attn_impl="eager"):
    """TF:dinov2/modeling_dinov2.py:342-380 (Dinov2Layer), :182-234 (attention), :281-297 (MLP),
    :272-278 (LayerScale).  Pre-LN:  h += l1*O(Attn(LN1 h));  h += l2*FC2(GELU(FC1(LN2 h)))."""
    bsz, n, d = h.shape
    dh = d // n_heads
    x = layer_norm(h, P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], eps)

    def proj(name):
        y = x @ P[f"{prefix}.attention.attention.{name}.weight"].t() + P[f"{prefix}.attention.attention.{name}.bias"]
        return y.view(bsz, n, n_heads, dh).transpose(1, 2)

    q, k, v = proj("query"), proj("key"), proj("value")
    ctx = attention(q, k, v, dh ** -0.5, None, attn_impl).transpose(1, 2).reshape(bsz, n, d)
    a = ctx @ P[f"{prefix}.attention.output.dense.weight"].t() + P[f"{prefix}.attention.output.dense.bias"]
    h = h + a * P[f"{prefix}.layer_scale1.lambda1"]
    x = layer_norm(h, P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], eps)
    x = gelu_erf(x @ P[f"{prefix}.mlp.fc1.weight"].t() + P[f"{prefix}.mlp.fc1.bias"])
    x = x @ P[f"{prefix}.mlp.fc2.weight"].t() + P[f"{prefix}.mlp.fc2.bias"]
    return h + x * P[f"{prefix}.layer_scale2.lambda1"]


# --------------------------------------------------------------------------------------------
# text: MPNet
# --------------------------------------------------------------------------------------------
def mpnet_position_ids(input_ids, padding_idx=1):
    """TF:mpnet/modeling_mpnet.py:873-881 (create_position_ids_from_input_ids)."""
    mask = (input_ids != padding_idx).to(torch.int64)
    return torch.cumsum(mask, dim=1) * mask + padding_idx


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """TF:mpnet/modeling_mpnet.py:331-348.  rel = memory_pos - context_pos (int64 tensor)."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).to(torch.int64) * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    # same fp32 expression as the reference so that bucket boundaries agree bit-for-bit
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.int64)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def relative_position_bucket_table(seq_len, num_buckets=32, max_distance=128):
    """(L, L) int64 bucket index for query i (row) / key j (col): TF:mpnet/modeling_mpnet.py:312-329."""
    ctx = torch.arange(seq_len, dtype=torch.int64)[:, None]
    mem = torch.arange(seq_len, dtype=torch.int64)[None, :]
    return relative_position_bucket(mem - ctx, num_buckets, max_distance)


def mpnet_forward(input_ids, attention_mask, P, n_layers, n_heads, eps, attn_impl="eager", prefix="text_model"):
    """TF:mpnet/modeling_mpnet.py:58-95 (embeddings), :264-348 (encoder), :115-231 (post-LN layer)."""
    t, l = input_ids.shape
    d = P[f"{prefix}.embeddings.word_embeddings.weight"].shape[1]
    dh = d // n_heads
    pos_ids = mpnet_position_ids(input_ids)
    h = P[f"{prefix}.embeddings.word_embeddings.weight"][input_ids] + P[f"{prefix}.embeddings.position_embeddings.weight"][pos_ids]
    h = layer_norm(h, P[f"{prefix}.embeddings.LayerNorm.weight"], P[f"{prefix}.embeddings.LayerNorm.bias"], eps)
    # relative position bias, shared by all layers: (1, H, L, L)
    buckets = relative_position_bucket_table(l, P[f"{prefix}.encoder.relative_attention_bias.weight"].shape[0])
    bias = P[f"{prefix}.encoder.relative_attention_bias.weight"][buckets].permute(2, 0, 1).unsqueeze(0)
    # key-padding mask: additive, most-negative finite value (HF extended / bidirectional mask)
    neg = torch.finfo(torch.float32).min
    kmask = torch.where(attention_mask[:, None, None, :] != 0, 0.0, neg).to(torch.float32)
    bias = bias + kmask
    for i in range(n_layers):
        pre = f"{prefix}.encoder.layer.{i}"

        def proj(name):
            y = h @ P[f"{pre}.attention.attn.{name}.weight"].t() + P[f"{pre}.attention.attn.{name}.bias"]
            return y.view(t, l, n_heads, dh).transpose(1, 2)

        ctx = attention(proj("q"), proj("k"), proj("v"), 1.0 / math.sqrt(dh), bias, attn_impl)
        ctx = ctx.transpose(1, 2).reshape(t, l, d)
        o = ctx @ P[f"{pre}.attention.attn.o.weight"].t() + P[f"{pre}.attention.attn.o.bias"]
        a = layer_norm(o + h, P[f"{pre}.attention.LayerNorm.weight"], P[f"{pre}.attention.LayerNorm.bias"], eps)
        x = gelu_erf(a @ P[f"{pre}.intermediate.dense.weight"].t() + P[f"{pre}.intermediate.dense.bias"])
        x = x @ P[f"{pre}.output.dense.weight"].t() + P[f"{pre}.output.dense.bias"]
        h = layer_norm(x + a, P[f"{pre}.output.LayerNorm.weight"], P[f"{pre}.output.LayerNorm.bias"], eps)
    return h


def masked_mean_pool(token_embeddings, attention_mask):
    """exp/cxr_pt/model/modeling.py:148-156: sum(h*m) / clamp(sum(m), 1e-9)."""
    m = attention_mask.unsqueeze(-1).to(token_embeddings.dtype)
    return (token_embeddings * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)


# --------------------------------------------------------------------------------------------
# VL-CABS head
# --------------------------------------------------------------------------------------------
def vlcabs(text_features, vision_tokens, ln_w, ln_b, ln_eps, temperature, sim_op="cos"):
    """exp/cxr_pt/model/losses.py:71-105 (shared LN on both sides, CLS kept) + :187-240 (SimilarityLogit).
    text_features (T, D) pre-LN; vision_tokens (B, N, D) pre-LN; temperature: the score temperature of sim_op 'cos'
    (exp(attn_temperature) if the checkpoint has one, else exp(loss_temperature): losses.py:175-181).
    Returns t2i_logits (T, B) (before the reference's .squeeze()) and pre-softmax scores (B, T, N)."""
    q = layer_norm(text_features, ln_w, ln_b, ln_eps)          # losses.py:163-164
    v = layer_norm(vision_tokens, ln_w, ln_b, ln_eps)          # losses.py:90-91
    if sim_op == "dot":                                        # losses.py:214-215: no normalisation, denominator sqrt(D)
        scores = torch.einsum("td,bnd->btn", q, v) / math.sqrt(v.shape[-1])
        p = torch.softmax(scores, dim=-1)
        agg = l2_normalize(torch.einsum("btn,bnd->btd", p, v))  # losses.py:224, :227
        logits_bt = torch.einsum("td,btd->bt", l2_normalize(q), agg)   # losses.py:226, :229-231
        return logits_bt.t().contiguous(), scores
    if sim_op != "cos":
        raise NotImplementedError                               # losses.py:216-217
    qn = l2_normalize(q)                                       # losses.py:212
    vn = l2_normalize(v)                                       # losses.py:213
    scores = torch.einsum("td,bnd->btn", qn, vn) / temperature  # losses.py:219-221
    p = torch.softmax(scores, dim=-1)                          # losses.py:222
    agg = torch.einsum("btn,bnd->btd", p, vn)                  # losses.py:224
    agg = l2_normalize(agg)                                    # losses.py:227
    logits_bt = torch.einsum("td,btd->bt", qn, agg)            # losses.py:229-231
    return logits_bt.t().contiguous(), scores                  # losses.py:233 (.T)


def interpolate_similarity_scores(similarity_scores, origin_size, keep_aspect_ratio=False):
    """exp/cxr_pt/inference/segmentation_utils.py:36-70: (g*g,) -> view (1,1,g,g) -> bilinear, align_corners=False
    -> (1, H, W).  keep_aspect_ratio=False: BlipImageProcessor branch (:62-70).  True: AspectRatioBlipImageProcessor
    branch (:41-60) — upsample to the padded square max(H, W), crop [pad_top:pad_top+H, pad_left:pad_left+W].
    Pinned by tests/golden/post_maps.npz (outputs of the reference's own function)."""
    h, w = origin_size
    g = int(similarity_scores.shape[-1] ** 0.5)
    s = similarity_scores.reshape(1, 1, g, g)
    if not keep_aspect_ratio:
        return F.interpolate(s, size=(h, w), mode="bilinear", align_corners=False).squeeze(1)
    p = max(h, w)
    sq = F.interpolate(s, size=(p, p), mode="bilinear", align_corners=False)
    left, top = (p - w) // 2, (p - h) // 2
    return sq[:, :, top:top + h, left:left + w].contiguous().squeeze(1)


def get_grounding_point(similarity_score, image_size, keep_aspect_ratio=False):
    """exp/cxr_pt/inference/grounding_utils.py:166-261: upsample as above, first flat maximum, unravel -> (x, y)."""
    h, w = image_size
    m = interpolate_similarity_scores(similarity_score, image_size, keep_aspect_ratio).reshape(-1)
    idx = int(torch.argmax(m))                   # torch.max(dim=0) on the flat map returns the first maximum
    return idx % w, idx // w


# --------------------------------------------------------------------------------------------
# the model object
# --------------------------------------------------------------------------------------------
class OracleModel:
    """Restatement of CxrAlignModel's inference methods (exp/cxr_pt/model/modeling.py:96-156, :278-328)."""

    def __init__(self, state_dict, cfg, attn_impl="eager"):
        self.cfg = cfg
        self.P = {k: _t(v).float() for k, v in state_dict.items()}
        self.attn_impl = attn_impl
        self._pos_cache = {}

    # modeling.py:96-123
    def vision_embeddings(self, pixel_values):
        P, cfg = self.P, self.cfg
        px = _t(pixel_values).float()
        gh, gw = px.shape[2] // cfg.patch_size, px.shape[3] // cfg.patch_size
        key = (gh, gw)
        if key not in self._pos_cache:
            self._pos_cache[key] = interpolate_pos_encoding(P["vision_model.embeddings.position_embeddings"], gh, gw)
        return patch_embed(px, P["vision_model.embeddings.patch_embeddings.projection.weight"],
                           P["vision_model.embeddings.patch_embeddings.projection.bias"],
                           P["vision_model.embeddings.cls_token"], self._pos_cache[key])

    def forward_vision_model(self, pixel_values, return_stages=False):
        P, cfg = self.P, self.cfg
        stages = {}
        h = self.vision_embeddings(pixel_values)
        stages["embeddings"] = h
        for i in range(cfg.vit_layers):
            h = dino_block(h, P, f"vision_model.encoder.layer.{i}", cfg.num_attention_heads,
                           cfg.vit_layer_norm_eps, self.attn_impl)
            stages[f"vit_layer_{i}"] = h
        h = layer_norm(h, P["vision_model.layernorm.weight"], P["vision_model.layernorm.bias"], cfg.vit_layer_norm_eps)
        stages["vit_final_ln"] = h
        for i in range(cfg.align_layers):
            h = dino_block(h, P, f"align_transformer.transformer_layers.layer.{i}", cfg.num_attention_heads,
                           cfg.vit_layer_norm_eps, self.attn_impl)
            stages[f"align_layer_{i}"] = h
        cls_token, patch_tokens = h[:, 0], h[:, 1:]
        image_features = l2_normalize(torch.cat([cls_token, patch_tokens.mean(dim=1)], dim=1))
        out = {"vision_tokens": h, "image_cls_token": cls_token, "image_patch_tokens": patch_tokens,
               "image_features": image_features}
        if return_stages:
            out["stages"] = stages
        return out

    # modeling.py:125-211 (MPNet branch)
    def forward_text_model(self, encoded_input):
        cfg = self.cfg
        ids = _t(encoded_input["input_ids"]).long()
        mask = _t(encoded_input["attention_mask"]).long()
        tok = mpnet_forward(ids, mask, self.P, cfg.text_layers, cfg.num_attention_heads,
                            cfg.text_layer_norm_eps, self.attn_impl)
        feat = masked_mean_pool(tok, mask)
        if "text_projector.weight" in self.P:                   # modeling.py:70-73, :199-200: nn.Linear(text_dim, 2 * hidden)
            feat = feat @ self.P["text_projector.weight"].T + self.P["text_projector.bias"]
        return {"text_features_wo_l2_norm": feat, "text_features": l2_normalize(feat)}

    def text_features(self, encoded, split_rows=True):
        """losses.py:126-166 via modeling.py:290-298: each prompt row is encoded on its own (keeping its pads)."""
        ids = _t(encoded["input_ids"]).long()
        mask = _t(encoded["attention_mask"]).long()
        if split_rows:
            feats = [self.forward_text_model({"input_ids": ids[i:i + 1], "attention_mask": mask[i:i + 1]})
                     ["text_features_wo_l2_norm"] for i in range(ids.shape[0])]
            return torch.cat(feats, dim=0)
        return self.forward_text_model({"input_ids": ids, "attention_mask": mask})["text_features_wo_l2_norm"]

    # modeling.py:278-328, compute_logits_type == "radzero"
    def compute_logits(self, pixel_values, encoded_key_phrases, text_features=None, **kwargs):
        P, cfg = self.P, self.cfg
        kind = getattr(cfg, "compute_logits_type", "radzero")
        if kind in ("cls_alignment", "global_alignment"):       # modeling.py:330-353: the groups' features concatenated, whole groups encoded at once
            vo = self.forward_vision_model(pixel_values)
            key = torch.cat([self.forward_text_model(kp)["text_features"] for kp in encoded_key_phrases], dim=0)
            if kind == "cls_alignment":
                return {"logits": vo["image_cls_token"] @ key.T}
            sim = torch.einsum("ind,jd->ijn", vo["image_patch_tokens"], key[:, cfg.hidden_size:])
            return {"logits": vo["image_features"] @ key.T, "similarity_scores": sim}
        vt = self.forward_vision_model(pixel_values)["vision_tokens"]
        if text_features is None:
            text_features = self.text_features(encoded_key_phrases[0])
        key = "loss_fns.RadZeroLoss.attn_temperature"           # losses.py:175-181: the attention temperature wins when it exists
        tau = float(torch.exp(P[key if key in P else "loss_fns.RadZeroLoss.loss_temperature"])[0])
        t2i, scores = vlcabs(text_features, vt, P["loss_fns.RadZeroLoss.layer_norm.weight"],
                             P["loss_fns.RadZeroLoss.layer_norm.bias"], cfg.shared_layer_norm_eps, tau, cfg.sim_op)
        t2i = t2i.squeeze()                                     # losses.py:229-231 .squeeze() degeneracy
        sim = scores[:, :, 1:] if cfg.use_vision_cls_token else scores   # modeling.py:311-317
        # modeling.py:322-328: divides by the (1,)-shaped parameter .exp() -> a 0-d t2i becomes (1,)
        logits = (t2i.T if t2i.dim() == 2 else t2i) / torch.exp(P["loss_fns.RadZeroLoss.loss_temperature"])
        return {"logits": logits, "similarity_scores": sim, "t2i_logits": t2i, "t2i_attn_weights": [scores]}
