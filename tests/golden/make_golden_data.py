"""Generate tests/golden/data_ref.pt: the reference's own `preprocess` / image-token clean-up / collate
(VisualRWKV-v7/v7.00/src/dataset.py) on records of its dummy_data/dummy.json with its own tokenizer
(tokenizer/rwkv_tokenizer.py + rwkv_vocab_v20230424.txt).  Stored: the JSON records used (data, from the reference's
dummy file), and for each the token ids of every turn text as the tokenizer produced them (so that the test needs no
vocabulary file), input_ids, labels, input_text.

Run where /root/reference exists:   python tests/golden/make_golden_data.py"""
import copy
import json
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/VisualRWKV-v7/v7.00"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "tokenizer"))


def main():
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_info = lambda *a, **k: None
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu})
    from src import dataset as ref
    from rwkv_tokenizer import TRIE_TOKENIZER
    tok = TRIE_TOKENIZER(os.path.join(REF, "tokenizer", "rwkv_vocab_v20230424.txt"))

    class Recording:
        """Wraps the tokenizer and remembers text -> ids, so that the test can replay it without the vocabulary."""
        def __init__(self):
            self.table = {}

        def encode(self, s):
            ids = tok.encode(s)
            self.table[s] = list(ids)
            return ids

    rec = Recording()
    data = json.load(open(os.path.join(REF, "dummy_data", "dummy.json")))
    picks = [data[i] for i in (0, 1, 2, 7, 40, 123, 500, 999)]
    # two synthetic variants exercising the other branches: no image, and an inference-style empty assistant turn
    noimg = {"id": "noimg", "conversations": [{"from": "human", "value": "  Hello\n\n\n there "}, {"from": "gpt", "value": "Hi.\n \nBye"}]}
    infer = {"id": "infer", "image": "x.jpg", "conversations": [{"from": "human", "value": "What is this?\n<image>"}, {"from": "gpt", "value": ""}]}
    picks += [noimg, infer]
    out = {"records": picks, "cases": []}
    for ctx_len, ntok in ((256, 16), (2048, 576)):
        for s in picks:
            if "image" in s:
                n = 1 if isinstance(s["image"], str) else len(s["image"])
                conv = ref.process_image_tokens_in_conversations(copy.deepcopy(s["conversations"]), num_image_paths=n)
            else:
                conv = ref.process_tokens_in_conversations(copy.deepcopy(s["conversations"]))
            cleaned = copy.deepcopy(conv)
            d = ref.preprocess(conv, rec, has_image="image" in s, ctx_len=ctx_len, num_token_per_image=ntok, pad_token_id=0)
            out["cases"].append({"id": s["id"], "ctx_len": ctx_len, "num_token_per_image": ntok, "cleaned": cleaned,
                                 "input_ids": d["input_ids"], "labels": d["labels"], "input_text": d["input_text"]})
    out["token_table"] = rec.table
    # collate of three prepared samples with dummy pixel tensors
    samples = []
    for i, s in enumerate(picks[:3]):
        conv = ref.process_image_tokens_in_conversations(copy.deepcopy(s["conversations"]), num_image_paths=1)
        d = ref.preprocess(conv, rec, has_image=True, ctx_len=64, num_token_per_image=4)
        d["images"] = {k: torch.full((1, 3, 2, 2), float(i)) for k in ("dino", "siglip", "sam")}
        d["sample_id"] = s["id"]
        samples.append(d)
    col = ref.multi_image_collate_fn(samples)
    out["collate"] = {"input_ids": col["input_ids"], "labels": col["labels"], "sample_id": col["sample_id"],
                      "num_image_per_sample": col["images"]["num_image_per_sample"], "dino": col["images"]["dino"]}
    path = os.path.join(HERE, "data_ref.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; token table entries:", len(rec.table))


if __name__ == "__main__":
    main()
