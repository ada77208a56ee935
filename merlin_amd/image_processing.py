"""Image preprocessing of the data path: `BaseDataset.image_processor` (mmgpt/data/dataset/base_dataset.py:178-197;
`process_image` in mmgpt/utils/mm_utils.py:29-45 is the eval-side twin) over transformers' `CLIPImageProcessor.preprocess`
(third-party; the PIL backend: convert-RGB -> resize shortest edge, bicubic -> center crop -> rescale 1/255 -> normalize).

Host-side CPU code (PIL + numpy), like the reference's DataLoader workers: it produces the float32 [3, H, W] tensors the
collator stacks into `images` (collator.py:29-34), which `mh_im2col_patches` then reads on the device.  Pinned by goldens
produced by the reference's own method on synthetic images (oracle/make_image_golden.py -> tests/golden/image_proc.npz).

Modes (`multimodal_cfg['image_aspect_ratio']`; pretrain.sh:38 uses `resize`):
  resize   image.resize((S, S)) [PIL default = bicubic], then rescale + normalize only
  pad      expand2square with the mean colour (top-left anchored paste), shortest edge -> S, no crop
  keep     shortest edge -> min(2S / aspect, S), no crop (variable size)
  (other)  the processor's defaults: shortest edge -> its size, center crop to its crop_size
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def expand2square(pil_img, background_color):
    """mm_utils.py:10-22: pad to a square with `background_color`, image pasted at the top-left corner."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    out = Image.new(pil_img.mode, (side, side), background_color)
    out.paste(pil_img)
    return out


class CLIPImageProcessor:
    """The subset of transformers.CLIPImageProcessor the reference's data path uses: attributes `crop_size`, `size`,
    `image_mean`, `image_std`, and `preprocess(image, return_tensors='pt', do_resize=..., do_center_crop=..., size=...)`
    returning {'pixel_values': [Tensor[3, H, W]]}."""

    def __init__(self, size=336, crop_size=None, image_mean=None, image_std=None):
        self.size = {"shortest_edge": size}
        c = size if crop_size is None else crop_size
        self.crop_size = {"height": c, "width": c}
        self.image_mean = list(OPENAI_CLIP_MEAN if image_mean is None else image_mean)
        self.image_std = list(OPENAI_CLIP_STD if image_std is None else image_std)

    @staticmethod
    def _resize_shortest(img, short):
        w, h = img.size
        s, l = (w, h) if w <= h else (h, w)
        new_s, new_l = short, int(short * l / s)  # transformers get_resize_output_image_size(default_to_square=False)
        nw, nh = (new_s, new_l) if w <= h else (new_l, new_s)
        return img.resize((nw, nh), resample=Image.BICUBIC)

    @staticmethod
    def _center_crop(arr, ch, cw):
        """arr [H, W, 3]; transformers center_crop: zero-pads when the image is smaller than the crop."""
        h, w = arr.shape[:2]
        top, left = (h - ch) // 2, (w - cw) // 2
        if top >= 0 and left >= 0:
            return arr[top: top + ch, left: left + cw]
        nh, nw = max(ch, h), max(cw, w)
        pad = np.zeros((nh, nw, arr.shape[2]), dtype=arr.dtype)
        pt, pl = int(np.ceil((nh - h) / 2)), int(np.ceil((nw - w) / 2))
        pad[pt: pt + h, pl: pl + w] = arr
        top, left = top + pt, left + pl
        return pad[max(0, top): max(0, top) + ch, max(0, left): max(0, left) + cw]

    def preprocess(self, image, return_tensors="pt", do_resize=True, do_center_crop=True, size=None, **kw):
        if not isinstance(image, Image.Image):
            image = Image.fromarray(np.asarray(image).astype(np.uint8))
        image = image.convert("RGB")
        if do_resize:
            image = self._resize_shortest(image, (size or self.size)["shortest_edge"])
        arr = np.asarray(image)
        if do_center_crop:
            arr = self._center_crop(arr, self.crop_size["height"], self.crop_size["width"])
        x = arr.astype(np.float32) * np.float32(1.0 / 255.0)                      # do_rescale
        x = (x - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)  # do_normalize
        return {"pixel_values": [torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))]}

    __call__ = preprocess


def _short_edge_keeping_aspect(size_wh, image_size):
    """'keep' mode: the short edge is image_size unless that would push the long edge past 2 x image_size.  The quotient is
    formed as (2 n) / (long / short) like mm_utils.py:31-34 so that the int() truncation lands on the same pixel count."""
    long_side, short_side = max(size_wh), min(size_wh)
    return int(min((2 * image_size) / (long_side / short_side), image_size))


def _plan(image, processor, image_size, mode):
    """mode -> (image handed to the CLIP processor, processor overrides); unknown modes fall through to the processor's own
    resize + centre crop (mm_utils.py:43-44)."""
    no_crop = dict(do_center_crop=False)
    if mode == "resize":   # squash to a square, nothing left for the processor to resample
        return image.resize((image_size, image_size)), dict(no_crop, do_resize=False)
    if mode == "pad":      # letterbox with the processor's mean colour, then resize the square
        fill = tuple(int(c * 255) for c in processor.image_mean)
        return expand2square(image, fill), dict(no_crop, size={"shortest_edge": image_size})
    if mode == "keep":     # aspect ratio kept, long edge bounded
        return image, dict(no_crop, size={"shortest_edge": _short_edge_keeping_aspect(image.size, image_size)})
    return image, {}


def process_image(image, processor, image_size, mode="resize"):
    """`BaseDataset.image_processor` (base_dataset.py:178-197 -> mm_utils.py:29-45): PIL image -> float32 [3, H, W]."""
    image, overrides = _plan(image, processor, image_size, mode)
    return processor.preprocess(image, return_tensors="pt", **overrides)["pixel_values"][0]
