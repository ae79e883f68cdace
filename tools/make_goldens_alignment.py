"""Generate tests/golden/g12_cls_alignment_*.npz / g13_global_alignment_*.npz by RUNNING THE REFERENCE (tools/ref_loader.py) with
compute_logits_type "cls_alignment" / "global_alignment" (exp/cxr_pt/model/modeling.py:330-353; the second needs text_config.use_text_projection,
modeling.py:70-73) on the synthetic checkpoint.  Build container only:  python tools/make_goldens_alignment.py
The key phrases come as TWO encoded groups of different padded length: the reference concatenates the groups' features (modeling.py:332-337)."""
from __future__ import annotations

import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.synthetic import synthetic_pixels, synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict, state_dict_digest  # noqa: E402
from ref_loader import load_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    torch.set_num_threads(8)
    seed = 20260103
    for name, over in (("g12_cls_alignment_s224_b2_t5", dict(compute_logits_type="cls_alignment")),
                       ("g13_global_alignment_s266_b2_t5", dict(compute_logits_type="global_alignment", use_text_projection=True))):
        cfg = dataclasses.replace(RadZeroConfig(), **over)
        sd = make_state_dict(cfg, seed)
        model = load_reference_model(cfg, sd, attn_implementation="eager")
        side = 266 if "s266" in name else 224
        px = synthetic_pixels(2, side, 1250)
        ids_a, mask_a = synthetic_prompts(3, 5, 9, 4340)
        ids_b, mask_b = synthetic_prompts(2, 8, 14, 4341)
        encs = [{"input_ids": torch.from_numpy(i), "attention_mask": torch.from_numpy(m)} for i, m in ((ids_a, mask_a), (ids_b, mask_b))]
        with torch.no_grad():
            out = model.compute_logits(torch.from_numpy(px), encs)
            vo = model.forward_vision_model(torch.from_numpy(px))
            tf = model.forward_text_model(encs[0])
        rec = dict(side=side, batch=2, px_seed=1250, input_ids_a=ids_a, attention_mask_a=mask_a, input_ids_b=ids_b, attention_mask_b=mask_b,
                   logits=out["logits"].numpy(), image_features=vo["image_features"].numpy(), text_features_a=tf["text_features"].numpy(),
                   text_features_wo_l2_norm_a=tf["text_features_wo_l2_norm"].numpy(), weights_seed=seed, weights_digest=state_dict_digest(sd),
                   torch_version=torch.__version__, compute_logits_type=cfg.compute_logits_type, use_text_projection=cfg.use_text_projection)
        if "similarity_scores" in out:
            rec["similarity_scores"] = out["similarity_scores"].numpy()
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)
        print(name, {k: tuple(v.shape) for k, v in rec.items() if hasattr(v, "shape") and v.ndim > 0 and k in ("logits", "similarity_scores", "image_features", "text_features_a")},
              "keys:", sorted(out.keys()), flush=True)


if __name__ == "__main__":
    main()
