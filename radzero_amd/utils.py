"""`model_inference` — the surface RadZero's README advertises (README.md:66, :104-111):

    similarity_prob, similarity_map = model_inference(image_path, "There is fibrosis",
                                                      tokenizer=..., image_processor=..., model=...)

Its body ships only on the HF Hub (trust_remote_code), not in the reference repository; it is reproduced
from the in-repo analogue `extract_similarity_map` (exp/cxr_pt/inference/visualization/attention_map_base.py:12-42)
+ `interpolate_similarity_scores` (exp/cxr_pt/inference/segmentation_utils.py:36-70, BlipImageProcessor branch).
ASSUMPTION (unpinned): `similarity_prob` = sigmoid(logits) — the reference's callers apply
`torch.sigmoid` to both maps and scores (attention_map_base.py:57, segmentation_utils.py:225); `similarity_map`
is returned un-squashed at the ORIGINAL image size (H, W), as the README prints its min/max.
"""
from __future__ import annotations

import numpy as np
import torch


def _to_pixel_tensor(processor_outputs) -> torch.Tensor:
    pv = processor_outputs["pixel_values"] if not torch.is_tensor(processor_outputs) else processor_outputs
    if not torch.is_tensor(pv):
        pv = torch.from_numpy(np.array(pv))
    pv = pv.float()
    return pv if pv.dim() == 4 else pv.unsqueeze(0)


@torch.no_grad()
def extract_similarity_map(image_path, text, model, image_processor, tokenizer):
    """Same signature and semantics as the reference helper (attention_map_base.py:12-42)."""
    from PIL import Image
    image = Image.open(image_path)
    width, height = image.size
    pixel_values = _to_pixel_tensor(image_processor(image)).to(model.device)
    tokenized = tokenizer(text, padding=True, truncation=True, return_tensors="pt").to(model.device)
    output = model.compute_logits(pixel_values, [tokenized])
    scores = output["similarity_scores"].reshape(-1)                 # assumes one image, one prompt
    sim_map = model.upsample_similarity(scores, (height, width))     # (H, W)
    return sim_map, output


@torch.no_grad()
def model_inference(image_path, text, tokenizer, image_processor, model):
    sim_map, output = extract_similarity_map(image_path, text, model, image_processor, tokenizer)
    similarity_prob = torch.sigmoid(output["logits"].float())
    return similarity_prob, sim_map
