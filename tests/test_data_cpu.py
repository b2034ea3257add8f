"""Host-side batch construction (visualrwkv_amd/data.py) against the reference's own `preprocess` / collate run on its
dummy_data records with its tokenizer (tests/golden/make_golden_data.py).  The fixture carries the tokenizer's output
for every text chunk, so no vocabulary file is needed here."""
import copy
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_ref.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


class ReplayTokenizer:
    def __init__(self, table):
        self.table = table

    def encode(self, s):
        return list(self.table[s])          # KeyError = the mirror split or cleaned the text differently


def test_preprocess_matches_reference(gold):
    from visualrwkv_amd import data
    tok = ReplayTokenizer(gold["token_table"])
    recs = gold["records"]                     # ids repeat in the dummy file: cases are stored record by record
    assert len(gold["cases"]) == 2 * len(recs)
    for n, case in enumerate(gold["cases"]):
        s = recs[n % len(recs)]
        assert s["id"] == case["id"]
        if "image" in s:
            conv = data.process_image_tokens_in_conversations(copy.deepcopy(s["conversations"]), num_image_paths=1)
        else:
            conv = data.process_tokens_in_conversations(copy.deepcopy(s["conversations"]))
        assert conv == case["cleaned"], case["id"]
        out = data.build_sample(s, tok, case["ctx_len"], case["num_token_per_image"])
        assert out["input_text"] == case["input_text"]
        assert torch.equal(out["input_ids"], case["input_ids"]), case["id"]
        assert torch.equal(out["labels"], case["labels"]), case["id"]
        assert out["input_ids"].shape == (case["ctx_len"],) and out["sample_id"] == s["id"]
        if "image" in s:
            n_img_tok = int((out["input_ids"] == data.IMAGE_TOKEN_INDEX).sum())
            assert n_img_tok in (case["num_token_per_image"], min(case["num_token_per_image"], case["ctx_len"]))
            assert set(out["images"]) == {"dino", "siglip", "sam"} and out["images"]["sam"].shape == (1, 3, 1024, 1024)


def test_label_masking_rules(gold):
    """Human turns and the 3-token "Assistant:" prefix are ignored; padding is id 0 / label -100."""
    from visualrwkv_amd import data
    tok = ReplayTokenizer(gold["token_table"])
    s = next(r for r in gold["records"] if r["id"] == "noimg")
    out = data.build_sample(s, tok, 64, 16)
    ids, lab = out["input_ids"], out["labels"]
    user_text = next(k for k in gold["token_table"] if k.startswith("User: Hello"))
    asst_text = next(k for k in gold["token_table"] if k.startswith("Assistant: Hi."))
    assert out["input_text"] == user_text + asst_text and user_text.endswith("\n\n") and "\n\n" not in user_text[:-2]
    n_user, n_asst = len(tok.encode(user_text)), len(tok.encode(asst_text))
    assert bool((lab[:n_user + 3] == data.IGNORE_INDEX).all())
    assert torch.equal(lab[n_user + 3:n_user + n_asst], ids[n_user + 3:n_user + n_asst])
    assert bool((ids[n_user + n_asst:] == 0).all()) and bool((lab[n_user + n_asst:] == data.IGNORE_INDEX).all())


def test_collate_matches_reference(gold):
    from visualrwkv_amd import data
    tok = ReplayTokenizer(gold["token_table"])
    samples = []
    for i, s in enumerate(gold["records"][:3]):
        pix = {k: torch.full((1, 3, 2, 2), float(i)) for k in ("dino", "siglip", "sam")}
        samples.append(data.build_sample(s, tok, 64, 4, pixel_values=pix))
    col = data.multi_image_collate_fn(samples)
    ref = gold["collate"]
    assert torch.equal(col["input_ids"], ref["input_ids"]) and torch.equal(col["labels"], ref["labels"])
    assert col["sample_id"] == ref["sample_id"] and col["images"]["num_image_per_sample"] == ref["num_image_per_sample"]
    assert torch.equal(col["images"]["dino"], ref["dino"])


def test_mismatched_image_count_is_rejected():
    from visualrwkv_amd import data
    conv = [{"from": "human", "value": "<image>\n<image>\nhi"}, {"from": "gpt", "value": "x"}]
    with pytest.raises(AssertionError):
        data.process_image_tokens_in_conversations(conv, num_image_paths=1)
    with pytest.raises(ValueError):
        data.add_speaker_and_signal([{"from": "robot", "value": "x"}])
