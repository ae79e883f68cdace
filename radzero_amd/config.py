from __future__ import annotations
"""Hyper-parameters of the RadZero VL-CABS inference path.

Field names follow the reference's YAML / HF configs so a reference user finds the same knobs:
  exp/cxr_pt/configs/radzero.yaml:16-46          (vision/text/align/loss settings)
  exp/cxr_pt/model/configuration.py:91-104       (align transformer = Dinov2Config defaults)
  transformers Dinov2Config / MPNetConfig         (third-party, pinned 4.39.3 in requirements.txt:247)
"""
from dataclasses import dataclass, asdict


@dataclass
class RadZeroConfig:
    # --- vision encoder: Dinov2 (vision_encoders.py:28-29) ---
    hidden_size: int = 768
    num_attention_heads: int = 12
    mlp_ratio: int = 4
    patch_size: int = 14
    num_channels: int = 3
    pretrain_image_size: int = 224          # pos-embed grid = (224//14)^2 = 16x16 (dinov2-base-xray-224)
    vit_layers: int = 12
    vit_layer_norm_eps: float = 1e-6
    # how the stored position grid is resized to another resolution (one-time host op, modeling.interpolate_pos_encoding):
    # "size" = F.interpolate(size=(gh, gw)) — transformers >= 4.45 (and the 5.x the goldens were generated with);
    # "scale_factor_0p1" = the pinned transformers 4.39.3 form, scale_factor=((gh + 0.1)/g0, (gw + 0.1)/g0): source coordinates differ by
    # up to ~0.05 grid cells.  The reference pins 4.39.3 (requirements.txt:247); pick this to reproduce THAT environment.
    pos_embed_interpolation: str = "size"
    # --- align transformer (radzero.yaml:29-34): Dinov2Encoder, no final LN ---
    align_layers: int = 2
    # --- text encoder: MPNet (text_encoders.py:13-14) ---
    vocab_size: int = 30527
    max_position_embeddings: int = 514
    text_layers: int = 12
    text_intermediate_size: int = 3072
    text_layer_norm_eps: float = 1e-5
    relative_attention_num_buckets: int = 32
    pad_token_id: int = 1
    # --- RadZeroLoss / VL-CABS head (losses.py:35-69, radzero.yaml:36-46) ---
    loss_temperature: float = 0.07          # stored as log(0.07) in the checkpoint
    shared_layer_norm_eps: float = 1e-5     # nn.LayerNorm default (losses.py:51)
    use_vision_cls_token: bool = True
    sim_op: str = "cos"                     # "cos" (released radzero.yaml:44) | "dot" (RadZeroLoss's constructor default, losses.py:45, :214-215)
    attn_temperature: float | None = None   # losses.py:57-63: a separate temperature for the score softmax ("cos" only); None = the loss temperature
    # --- which logits compute_logits returns (modeling.py:288 / :330 / :340; the released radzero.yaml:48 says "radzero") ---
    compute_logits_type: str = "radzero"    # "radzero" (VL-CABS) | "cls_alignment" | "global_alignment"
    use_text_projection: bool = False       # text_projector = nn.Linear(768, 2 * 768) (modeling.py:70-73): what "global_alignment" needs

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def intermediate_size(self) -> int:
        return self.hidden_size * self.mlp_ratio

    @property
    def num_blocks(self) -> int:
        return self.vit_layers + self.align_layers

    def grid(self, image_side: int) -> int:
        return image_side // self.patch_size

    def tokens(self, image_side: int) -> int:
        g = self.grid(image_side)
        return g * g + 1

    def to_dict(self):
        return asdict(self)


def flops_per_image(cfg: RadZeroConfig, image_side: int, n_prompts: int) -> float:
    """Algorithmic FLOPs per image, SURVEY.md §8(d):
    F = blocks*(24*N*D^2 + 4*N^2*D) + 2*Np*588*D + 4*T*N*D."""
    d = cfg.hidden_size
    n = cfg.tokens(image_side)
    np_ = n - 1
    kpatch = cfg.num_channels * cfg.patch_size * cfg.patch_size
    return (cfg.num_blocks * (24.0 * n * d * d + 4.0 * n * n * d)
            + 2.0 * np_ * kpatch * d + 4.0 * n_prompts * n * d)


def attention_flops_per_image_layer(cfg: RadZeroConfig, image_side: int) -> float:
    n = cfg.tokens(image_side)
    return 4.0 * n * n * cfg.hidden_size
