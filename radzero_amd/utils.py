"""`model_inference` — the surface RadZero's README advertises (README.md:66, :104-111):

    similarity_prob, similarity_map = model_inference(image_path, "There is fibrosis",
                                                      tokenizer=..., image_processor=..., model=...)

Its body ships only on the HF Hub (trust_remote_code), not in the reference repository; it is reproduced
from the in-repo analogue `extract_similarity_map` (exp/cxr_pt/inference/visualization/attention_map_base.py:12-42)
+ `interpolate_similarity_scores` (exp/cxr_pt/inference/segmentation_utils.py:36-70).
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


def processor_keeps_aspect_ratio(image_processor) -> bool:
    """Which branch of the reference's map post-processing an image processor selects
    (segmentation_utils.py:41 / :62, grounding_utils.py:172 / :185: `isinstance` against AspectRatioBlipImageProcessor,
    then BlipImageProcessor).  Decided on the class names along the MRO so that neither `transformers` nor the reference's
    `processing.py` has to be importable here; `BlipImageProcessorPil` is what transformers >= 5 hands out when torchvision
    is absent.  Any other processor (Bit, M3AE, …) raises like grounding_utils.py:248-251: those branches belong to
    other backbones (SURVEY.md §2 OUT)."""
    names = [c.__name__ for c in type(image_processor).__mro__]
    if "AspectRatioBlipImageProcessor" in names:
        return True
    if any(n in ("BlipImageProcessor", "BlipImageProcessorPil", "BlipImageProcessorFast") for n in names):
        return False
    raise NotImplementedError(f"Image processor {type(image_processor)} is not supported")


@torch.no_grad()
def _similarity_map_and_output(image_path, text, model, image_processor, tokenizer):
    from PIL import Image
    keep_aspect = processor_keeps_aspect_ratio(image_processor)
    image = Image.open(image_path)
    width, height = image.size
    pixel_values = _to_pixel_tensor(image_processor(image)).to(model.device)
    tokenized = tokenizer(text, padding=True, truncation=True, return_tensors="pt").to(model.device)
    output = model.compute_logits(pixel_values, [tokenized])
    scores = output["similarity_scores"].reshape(-1)                 # one image, one prompt (attention_map_base.py:33-34)
    sim_map = model.upsample_similarity(scores, (height, width), keep_aspect_ratio=keep_aspect)     # (H, W)
    return sim_map, output


@torch.no_grad()
def extract_similarity_map(image_path, text, model, image_processor, tokenizer):
    """Same signature, return value and processor-dependent registration as the reference helper
    (attention_map_base.py:12-42): the (H, W) similarity map of the ORIGINAL image, un-squashed."""
    return _similarity_map_and_output(image_path, text, model, image_processor, tokenizer)[0]


@torch.no_grad()
def model_inference(image_path, text, tokenizer, image_processor, model):
    sim_map, output = _similarity_map_and_output(image_path, text, model, image_processor, tokenizer)
    similarity_prob = torch.sigmoid(output["logits"].float())
    return similarity_prob, sim_map
