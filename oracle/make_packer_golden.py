"""Golden vectors for the sequence-merge packers, produced by the REAL reference classes in the
build container: PairWebDataset / InterPairWebDataset / InterleaveWebDataset `token_processor`s and the collator are
called (object.__new__, no shard I/O) on scripted samples with the deterministic tests/toy_tokenizer.py.  I/O-only
third-party imports (webdataset, megfile, boto3, ...) are stubbed; the arithmetic is the reference's own.
Run here only (reads /root/reference); writes tests/golden/packers.json.   usage: python oracle/make_packer_golden.py"""
import json
import logging
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
import transformers  # noqa: F401,E402  (before the stubs)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


_stub("loguru", logger=logging.getLogger("ref"))
for n in ("megfile", "webdataset", "ipdb", "boto3", "smart_open", "timm", "timm.models", "timm.models.hub"):
    m = _stub(n)
    m.__getattr__ = lambda k: _Any()
sys.modules["megfile"].s3_path = _Any()
_stub("torchvision"); _stub("torchvision.transforms", InterpolationMode=_Any()); _stub("torchvision.transforms.functional", InterpolationMode=_Any())
const = types.ModuleType("mmgpt.utils.constants")
src = "".join(open("/root/reference/mmgpt/utils/constants.py").readlines()[:23])
exec(src, const.__dict__)
const.PAIR_WEBDATA = {}
const.INTERLEAVE_WEBDATA = {}
sys.path.insert(0, "/root/reference")
import mmgpt  # noqa: E402
import mmgpt.utils  # noqa: E402
sys.modules["mmgpt.utils.constants"] = const
mmgpt.utils.constants = const

from mmgpt.data.dataset.pair_webdataset import PairWebDataset  # noqa: E402
from mmgpt.data.dataset.interpair_webdataset import InterPairWebDataset  # noqa: E402
from mmgpt.data.dataset.interleave_webdataset import InterleaveWebDataset  # noqa: E402
from mmgpt.data.collator import DataCollatorForSupervisedDataset  # noqa: E402
from toy_tokenizer import ToyTokenizer  # noqa: E402

P = 4
IMG = 8


def make(cls, max_len):
    ds = object.__new__(cls)
    ds.tokenizer = ToyTokenizer(max_len)
    ds.multimodal_cfg = dict(image_token_len=P, use_im_start_end=True)
    ds.use_im_start_end = True
    ds.im_patch_token, ds.im_start_token, ds.im_end_token = 32000, 32001, 32002
    ds.image_size = IMG
    ds.replace_token = "<im_start>" + "<im_patch>" * P + "<im_end>"
    return ds


def imgs(n):
    return [torch.full((3, IMG, IMG), float(i + 1)) for i in range(n)]


def out(d):
    return dict(input_ids=d["input_ids"].tolist(), labels=d["labels"].tolist(), n_images=len(d["image"]),
                image_tags=[float(im[0, 0, 0]) for im in d["image"]])


cases = {"P": P, "image_size": IMG, "pair": [], "interpair": [], "interleave": [], "collate": []}
pair_samples = [
    (64, [("describe <image> briefly", "a cat on a mat"), (None, "two dogs run"), ("what is this", "a red car")]),
    (30, [("describe <image> briefly", "a cat on a mat"), (None, "two dogs run in the park near the river"), ("what is this", "a red car")]),
    (64, [(None, "<image>\nlone caption")]),
    (12, [("p <image>", "this answer is definitely too long to fit in twelve tokens"), (None, "x")]),
]
for max_len, tl in pair_samples:
    ds = make(PairWebDataset, max_len)
    cases["pair"].append(dict(max_len=max_len, text_list=tl, n_in=len(tl), out=out(ds.token_processor(tl, imgs(len(tl))))))
inter_samples = [
    (128, [("track <image> <image> the object", "it moves left"), ("detect <image>\n objects", "one box")], 3),
    (40, [("track <image> <image> the object", "it moves left"), ("detect <image>\n objects", "one box here and there"), ("more <image>", "dropped")], 4),
    (64, [("no image token here", "plain answer")], 1),
    (30, [("a <image> b <image> c <image>", "does not fit at all because it is a long answer text")], 3),
]
for max_len, tl, n in inter_samples:
    ds = make(InterPairWebDataset, max_len)
    cases["interpair"].append(dict(max_len=max_len, text_list=tl, n_in=n, out=out(ds.token_processor(tl, imgs(n)))))
leave_samples = [
    (128, ["first sentence", "second one", "third"], [0, 2]),
    (128, ["first sentence", "second one"], [1, 2]),      # image after the last sentence
    (128, ["only text here"], []),
    (20, ["a b c d", "e f g h", "i j k l"], [0, 1, 2]),    # truncation cuts through images
    (128, ["x y", "z"], [0, 5]),                           # out-of-range index dropped
]
for max_len, tl, idx in leave_samples:
    ds = make(InterleaveWebDataset, max_len)
    text = ds.multimodal_processor(tl, list(idx))
    d = ds.token_processor([text])
    d = dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
    # the tail of to_dict (interleave_webdataset.py:165-183) on already-decoded images
    image_list = imgs(len([i for i in idx]))
    lefts = torch.where(d["input_ids"] == 32001)[0]
    nr = 0
    if lefts.shape[0] > 0 and len(image_list) > 0:
        rights = lefts + P + 1
        nr = torch.where(rights < d["input_ids"].shape[0])[0].shape[0]
        if nr < lefts.shape[0]:
            d["input_ids"] = torch.cat([d["input_ids"][:lefts[nr]], torch.tensor([2])])
            d["labels"] = torch.cat([d["labels"][:lefts[nr]], torch.tensor([2])])
    d["image"] = image_list[:nr] if (nr > 0 and len(image_list) > 0) else [torch.zeros(3, IMG, IMG)]
    cases["interleave"].append(dict(max_len=max_len, text_list=tl, index_list=list(idx), text=text, out=out(d)))
# collator
tok = ToyTokenizer(24)
col = DataCollatorForSupervisedDataset(tokenizer=tok)
inst = []
for max_len, tl in pair_samples[:3]:
    inst.append(make(PairWebDataset, max_len).token_processor(tl, imgs(len(tl))))
b = col(inst)
cases["collate"].append(dict(model_max_length=24, from_pair_cases=[0, 1, 2], input_ids=b["input_ids"].tolist(), labels=b["labels"].tolist(),
                             attention_mask=b["attention_mask"].tolist(), image_shapes=[list(x.shape) for x in b["images"]]))
json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "packers.json"), "w"))
print({k: len(v) for k, v in cases.items() if isinstance(v, list)})
