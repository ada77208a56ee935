"""Sequence-merge packers of the training data path: the text side of
`PairWebDataset` / `InterPairWebDataset` / `InterleaveWebDataset` and the batch collator, restated without the
webdataset / megfile / S3 plumbing so that any sample source (tar shards, a list, a synthetic generator) can drive the
HIP path with exactly the reference's token and label layout.

What the packers guarantee (and the splice kernel relies on, base_mmgpt.py:116-135):
  * every image is `<im_start>` + P x `<im_patch>` + `<im_end>` (+ "\\n"), P = image_token_len;
  * prompts are masked (-100), answers (+ eos) are targets, all image-token positions are masked;
  * a merged sequence never exceeds `tokenizer.model_max_length`; pairs that do not fit are dropped TOGETHER WITH
    their images, so #images == #`<im_start>` in `input_ids`;
  * an empty image list becomes one all-zeros image (the model adds 0 * projector(dummy)).

`tokenizer` is anything with the Hugging Face call interface the reference uses: `tokenizer(text, max_length=...,
truncation=True).input_ids`, `.eos_token`, `.eos_token_id`, `.pad_token_id`, `.model_max_length`,
`.convert_tokens_to_ids`.  Pure host code (lists and CPU tensors); nothing here touches the GPU."""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

IGNORE_INDEX = -100                      # mmgpt/utils/constants.py:7-12
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


@dataclass
class PackerConfig:
    image_token_len: int                 # P = data_args.image_token_len (base_mmgpt.py:46)
    use_im_start_end: bool = True
    image_size: int = 336
    im_patch_token: int = 32000
    im_start_token: int = 32001
    im_end_token: int = 32002


class _PackerBase:
    def __init__(self, tokenizer, cfg: PackerConfig):
        self.tokenizer, self.cfg = tokenizer, cfg
        rt = DEFAULT_IM_PATCH_TOKEN * cfg.image_token_len
        self.replace_token = (DEFAULT_IM_START_TOKEN + rt + DEFAULT_IM_END_TOKEN) if cfg.use_im_start_end else rt

    def _mask_image_tokens(self, targets: torch.Tensor) -> torch.Tensor:
        m = targets == self.cfg.im_patch_token
        if self.cfg.use_im_start_end:
            m = m | (targets == self.cfg.im_start_token) | (targets == self.cfg.im_end_token)
        return targets.masked_fill(m, IGNORE_INDEX)

    def _dummy_image(self):
        return torch.zeros(3, self.cfg.image_size, self.cfg.image_size)

    def _tokenize_pair(self, prompt: Optional[str], text: str):
        """Prompt and answer are tokenized separately (prompt masked later); the answer gets eos and, when a prompt
        precedes it, loses its duplicate BOS (pair_webdataset.py:83-101 / interpair_webdataset.py:86-105)."""
        tok = self.tokenizer
        p_ids = list(tok(prompt, padding="longest", max_length=tok.model_max_length, truncation=True).input_ids) if prompt is not None else []
        t_ids = list(tok(text + tok.eos_token, padding="longest", max_length=tok.model_max_length - len(p_ids), truncation=True).input_ids)
        if prompt is not None and t_ids[0] == 1:
            t_ids = t_ids[1:]
        return p_ids, t_ids

    def _finish(self, input_ids, targets, image_list):
        input_ids = torch.tensor(input_ids, dtype=torch.long)
        targets = self._mask_image_tokens(torch.tensor(targets, dtype=torch.long))
        if len(image_list) == 0:
            image_list = [self._dummy_image()]
        return dict(image=image_list, input_ids=input_ids, labels=targets)


class PairPacker(_PackerBase):
    """`PairWebDataset.token_processor` (pair_webdataset.py:60-153): `merge_round` (prompt, caption) pairs, ONE image
    per pair, merged into one causal sequence."""

    def add_image_token(self, text: str) -> str:
        if DEFAULT_IMAGE_TOKEN in text:
            return text.replace(DEFAULT_IMAGE_TOKEN, self.replace_token)
        return self.replace_token + "\n" + text

    def __call__(self, text_list: Sequence[Tuple[Optional[str], str]], image_list: list):
        input_ids, targets = [], []
        for i, (prompt, text) in enumerate(text_list):
            if prompt is not None:
                prompt = self.add_image_token(prompt)
            else:
                text = self.add_image_token(text)
            p_ids, t_ids = self._tokenize_pair(prompt, text)
            if len(input_ids) + len(p_ids) + len(t_ids) > self.tokenizer.model_max_length:
                image_list = image_list[:i]  # one image per pair: drop the images of the dropped pairs
                break
            input_ids.extend(p_ids + t_ids)
            targets.extend([IGNORE_INDEX] * len(p_ids) + t_ids)
        return self._finish(input_ids, targets, image_list)


class InterPairPacker(_PackerBase):
    """`InterPairWebDataset.token_processor` (interpair_webdataset.py:53-162): pairs that may carry SEVERAL images each
    (multi-frame tracking / detection prompts); the image list is cut by the number of images actually consumed."""

    def add_image_token(self, text: str) -> str:
        if DEFAULT_IMAGE_TOKEN + "\n" in text:      # detection data: "<image>\n"
            return text.replace(DEFAULT_IMAGE_TOKEN, self.replace_token)
        if DEFAULT_IMAGE_TOKEN in text:             # tracking data: "<image>" without the newline
            return text.replace(DEFAULT_IMAGE_TOKEN, self.replace_token + "\n")
        return self.replace_token + "\n" + text

    def __call__(self, text_list: Sequence[Tuple[Optional[str], str]], image_list: list):
        img_count = 0
        input_ids, targets = [], []
        for i, (prompt, text) in enumerate(text_list):
            # (the reference evaluates prompt.count() before its None check: a None prompt with this dataset raises there
            # too; callers of this dataset always pass a prompt string)
            cur = prompt.count(DEFAULT_IMAGE_TOKEN) + text.count(DEFAULT_IMAGE_TOKEN)
            if cur == 0:
                cur = 1
            if prompt is not None:
                prompt = self.add_image_token(prompt)
            else:
                text = self.add_image_token(text)
            p_ids, t_ids = self._tokenize_pair(prompt, text)
            if len(input_ids) + len(p_ids) + len(t_ids) > self.tokenizer.model_max_length:
                image_list = image_list[:img_count]
                break
            input_ids.extend(p_ids + t_ids)
            targets.extend([IGNORE_INDEX] * len(p_ids) + t_ids)
            img_count += cur
        return self._finish(input_ids, targets, image_list)


class InterleavePacker(_PackerBase):
    """`InterleaveWebDataset` (interleave_webdataset.py:47-185): one document = sentences + images matched to sentence
    indices; every image goes in front of its sentence; images whose tokens do not fit are cut and eos is appended."""

    @staticmethod
    def select_images(image_infos: Sequence[dict], min_sim: float = 0.25):
        """Indices of the images kept by the CLIP-similarity filter (interleave_webdataset.py:127-138) and their
        `matched_text_index`: returns [(position in image_infos, text index)]."""
        keep = []
        for j, info in enumerate(image_infos):
            sim = info.get("matched_sim", info.get("match_sim", 1))
            if sim < min_sim:
                continue
            keep.append((j, info["matched_text_index"]))
        return keep

    def multimodal_text(self, text_list: Sequence[str], image_text_index_list: Sequence[int]) -> str:
        new = [copy.deepcopy(t) for t in text_list]
        idx = list(image_text_index_list)
        if len(idx) == 0:
            pass
        elif idx[-1] == len(new):
            new.append("")
        elif idx[-1] > len(new):
            while idx[-1] > len(new):
                idx = idx[:-1]
        for k in idx:
            new[k] = DEFAULT_IMAGE_TOKEN + "\n" + new[k]
        text = " ".join(new) + self.tokenizer.eos_token
        return text.replace(DEFAULT_IMAGE_TOKEN, self.replace_token)

    def tokens(self, text: str):
        tok = self.tokenizer
        input_ids = tok([text], return_tensors="pt", padding="longest", max_length=tok.model_max_length, truncation=True).input_ids
        targets = input_ids.clone()
        targets = targets.masked_fill((targets == tok.pad_token_id) | (targets == self.cfg.im_patch_token), IGNORE_INDEX)
        if self.cfg.use_im_start_end:
            targets = targets.masked_fill((targets == self.cfg.im_start_token) | (targets == self.cfg.im_end_token), IGNORE_INDEX)
        return input_ids[0], targets[0]

    def __call__(self, text_list: Sequence[str], image_list: list, image_text_index_list: Sequence[int]):
        tok = self.tokenizer
        input_ids, labels = self.tokens(self.multimodal_text(text_list, image_text_index_list))
        lefts = torch.where(input_ids == self.cfg.im_start_token)[0]
        n_right = 0
        if lefts.shape[0] > 0 and len(image_list) > 0:
            rights = lefts + self.cfg.image_token_len + 1
            n_right = int((rights < input_ids.shape[0]).sum())
            if n_right < lefts.shape[0]:  # the truncation cut through an image: drop it and everything after, close with eos
                cut = int(lefts[n_right])
                input_ids = torch.cat([input_ids[:cut], torch.tensor([tok.eos_token_id])])
                labels = torch.cat([labels[:cut], torch.tensor([tok.eos_token_id])])
        images = image_list[:n_right] if (n_right > 0 and len(image_list) > 0) else [self._dummy_image()]
        return dict(input_ids=input_ids, labels=labels, image=images)


def collate(instances: Sequence[dict], pad_token_id: int, model_max_length: int) -> dict:
    """`DataCollatorForSupervisedDataset.__call__` (collator.py:12-34): right-pad ids with pad, labels with -100, clip to
    the context length, attention_mask = ids != pad, images = one stacked [n_i, 3, H, W] tensor per sample."""
    input_ids = torch.nn.utils.rnn.pad_sequence([x["input_ids"] for x in instances], batch_first=True, padding_value=pad_token_id)
    labels = torch.nn.utils.rnn.pad_sequence([x["labels"] for x in instances], batch_first=True, padding_value=IGNORE_INDEX)
    input_ids, labels = input_ids[:, :model_max_length], labels[:, :model_max_length]
    return dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(pad_token_id),
                images=[torch.stack(x["image"]) for x in instances])


def splice_table(input_ids: torch.Tensor, images_per_sample: Sequence[int], P: int, im_start_token: int, im_end_token: int) -> torch.Tensor:
    """Host-side form of the index table the splice kernel (`mh_splice_index`) builds on the device: src[b, s] = row of
    the flattened image-feature matrix that replaces position s, or -1 (keep the token embedding).  Image k of sample b
    starts at feature row (sum of earlier samples' images + k) * P.  Raises ValueError on the reference's two checks
    (base_mmgpt.py:116-118,125-126); images beyond the number of `<im_start>` are ignored (zip, :121)."""
    B, S = input_ids.shape
    src = torch.full((B, S), -1, dtype=torch.int32)
    base = 0
    for b in range(B):
        starts = torch.where(input_ids[b] == im_start_token)[0].tolist()
        ends = torch.where(input_ids[b] == im_end_token)[0].tolist()
        if len(starts) != len(ends):
            raise ValueError("The number of image start tokens and image end tokens should be the same.")
        for k, p0 in enumerate(starts[: images_per_sample[b]]):
            if p0 + P + 1 >= S or int(input_ids[b, p0 + P + 1]) != im_end_token:
                raise ValueError("The image end token should follow the image start token.")
            src[b, p0 + 1: p0 + 1 + P] = torch.arange(P, dtype=torch.int32) + (base + k) * P
        base += images_per_sample[b]
    return src
