"""The build's sequence-merge packers (merlin_amd/packers.py) against golden vectors produced by the REAL reference
classes (oracle/make_packer_golden.py -> tests/golden/packers.json) on the same scripted samples and the same
deterministic tokenizer: token ids, labels (prompt / image-token masking), which images survive truncation, the
interleaved text, the collated batch; plus the host-side splice table against the rule of base_mmgpt.py:116-135."""
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from toy_tokenizer import ToyTokenizer  # noqa: E402

from merlin_amd import packers as PK  # noqa: E402

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "packers.json")))
P, IMG = G["P"], G["image_size"]


def _cfg():
    return PK.PackerConfig(image_token_len=P, use_im_start_end=True, image_size=IMG)


def _imgs(n):
    return [torch.full((3, IMG, IMG), float(i + 1)) for i in range(n)]


def _check(d, ref):
    assert d["input_ids"].tolist() == ref["input_ids"]
    assert d["labels"].tolist() == ref["labels"]
    assert len(d["image"]) == ref["n_images"]
    assert [float(im[0, 0, 0]) for im in d["image"]] == ref["image_tags"]
    assert d["input_ids"].dtype == torch.long and d["labels"].dtype == torch.long


@pytest.mark.parametrize("i", range(len(G["pair"])))
def test_pair_packer(i):
    c = G["pair"][i]
    tl = [tuple(x) for x in c["text_list"]]
    _check(PK.PairPacker(ToyTokenizer(c["max_len"]), _cfg())(tl, _imgs(c["n_in"])), c["out"])


@pytest.mark.parametrize("i", range(len(G["interpair"])))
def test_interpair_packer(i):
    c = G["interpair"][i]
    tl = [tuple(x) for x in c["text_list"]]
    d = PK.InterPairPacker(ToyTokenizer(c["max_len"]), _cfg())(tl, _imgs(c["n_in"]))
    _check(d, c["out"])
    # the invariant the splice relies on: one image per <im_start> (unless the list fell back to the dummy image)
    n_start = int((d["input_ids"] == 32001).sum())
    assert n_start == len(d["image"]) or (n_start == 0 and len(d["image"]) == 1)


@pytest.mark.parametrize("i", range(len(G["interleave"])))
def test_interleave_packer(i):
    c = G["interleave"][i]
    pk = PK.InterleavePacker(ToyTokenizer(c["max_len"]), _cfg())
    assert pk.multimodal_text(c["text_list"], c["index_list"]) == c["text"]
    _check(pk(c["text_list"], _imgs(len(c["index_list"])), c["index_list"]), c["out"])


def test_interleave_similarity_filter():
    infos = [dict(image_name="a", matched_sim=0.3, matched_text_index=0), dict(image_name="b", match_sim=0.1, matched_text_index=1),
             dict(image_name="c", matched_text_index=2)]
    assert PK.InterleavePacker.select_images(infos) == [(0, 0), (2, 2)]


def test_collate_matches_reference():
    c = G["collate"][0]
    inst = []
    for k in c["from_pair_cases"]:
        pc = G["pair"][k]
        inst.append(PK.PairPacker(ToyTokenizer(pc["max_len"]), _cfg())([tuple(x) for x in pc["text_list"]], _imgs(pc["n_in"])))
    b = PK.collate(inst, pad_token_id=0, model_max_length=c["model_max_length"])
    assert b["input_ids"].tolist() == c["input_ids"] and b["labels"].tolist() == c["labels"]
    assert b["attention_mask"].tolist() == c["attention_mask"] and b["attention_mask"].dtype == torch.bool
    assert [list(x.shape) for x in b["images"]] == c["image_shapes"]


def test_splice_table_rule():
    c = G["interpair"][0]
    d = PK.InterPairPacker(ToyTokenizer(c["max_len"]), _cfg())([tuple(x) for x in c["text_list"]], _imgs(c["n_in"]))
    ids = d["input_ids"][None]
    src = PK.splice_table(ids, [len(d["image"])], P, 32001, 32002)
    starts = torch.where(ids[0] == 32001)[0].tolist()
    assert len(starts) == 3
    for k, p0 in enumerate(starts):
        assert src[0, p0 + 1: p0 + 1 + P].tolist() == list(range(k * P, (k + 1) * P))
        assert int(src[0, p0]) == -1 and int(src[0, p0 + P + 1]) == -1
    assert int((src >= 0).sum()) == 3 * P
    bad = ids.clone(); bad[0, starts[0] + P + 1] = 5          # <im_end> not where it must be
    with pytest.raises(ValueError):
        PK.splice_table(bad, [3], P, 32001, 32002)
    # extra images beyond the <im_start> count are ignored (zip semantics)
    assert torch.equal(PK.splice_table(ids, [7], P, 32001, 32002), src)
