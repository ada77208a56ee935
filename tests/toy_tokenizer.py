"""A deterministic stand-in for the Llama tokenizer's INTERFACE (what the reference's packers call: __call__ with
max_length/truncation[/return_tensors], .eos_token, .eos_token_id, .pad_token_id, .model_max_length,
.convert_tokens_to_ids) so that the reference's token_processor code and the build's packers can be run on the same
inputs without a vocabulary file.  BOS (id 1) is prepended to every call like Llama's tokenizer does; the three image
tokens and </s> are single ids; "\n" is id 13; every other whitespace-separated word hashes to [3, 32000)."""
import re
import types
import zlib

import torch

SPECIAL = {"<im_patch>": 32000, "<im_start>": 32001, "<im_end>": 32002, "</s>": 2, "<unk>": 0}
_SPLIT = re.compile(r"(<im_patch>|<im_start>|<im_end>|</s>|\n)")


class ToyTokenizer:
    eos_token, eos_token_id, pad_token_id, bos_token_id = "</s>", 2, 0, 1

    def __init__(self, model_max_length=2048):
        self.model_max_length = model_max_length

    def _ids(self, text):
        ids = [1]
        for piece in _SPLIT.split(text):
            if piece in SPECIAL:
                ids.append(SPECIAL[piece])
            elif piece == "\n":
                ids.append(13)
            else:
                ids += [3 + zlib.crc32(w.encode()) % 31997 for w in piece.split()]
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=False):
        batch = isinstance(text, (list, tuple))
        rows = [self._ids(t) for t in (text if batch else [text])]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        if return_tensors == "pt":
            L = max(len(r) for r in rows)
            return types.SimpleNamespace(input_ids=torch.tensor([r + [self.pad_token_id] * (L - len(r)) for r in rows], dtype=torch.long))
        return types.SimpleNamespace(input_ids=rows if batch else rows[0])

    def convert_tokens_to_ids(self, toks):
        return [SPECIAL[t] for t in toks]
