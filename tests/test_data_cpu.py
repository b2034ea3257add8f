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


def _make_dataset(tmp_path, gold, missing=()):
    """The reference's dummy records with freshly written JPEGs (one per record name) under tmp_path."""
    import json
    from types import SimpleNamespace

    import numpy as np
    from PIL import Image
    recs = copy.deepcopy(gold["records"])
    rng = np.random.default_rng(0)
    for n, r in enumerate(recs):
        if "image" not in r:
            continue
        path = tmp_path / r["image"]
        path.parent.mkdir(parents=True, exist_ok=True)
        if r["image"] not in missing and not path.exists():
            h, w = 40 + 7 * (n % 5), 64 + 5 * (n % 3)
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path, quality=95)
    (tmp_path / "data.json").write_text(json.dumps(recs))
    case = gold["cases"][0]
    args = SimpleNamespace(data_file=str(tmp_path / "data.json"), image_folder=str(tmp_path), tokenizer=ReplayTokenizer(gold["token_table"]),
                           ctx_len=case["ctx_len"], num_token_per_image=case["num_token_per_image"], epoch_steps=3, real_bsz=4, micro_bsz=2,
                           vocab_size=65536)
    return recs, args


@pytest.mark.parametrize("workers", [0, 2])
def test_dataset_loader_prefetcher(tmp_path, gold, workers):
    """MyDataset -> DataLoader -> DevicePrefetcher (on the CPU here): the reference's rank-strided sampling order, its
    token ids / labels per record, and the three tower tensors produced after the hand-off (decode in the workers,
    transform after it) equal to transforming the decoded image directly."""
    import numpy as np
    from PIL import Image

    from visualrwkv_amd import data, dp, image
    recs, args = _make_dataset(tmp_path, gold)
    ds = data.MyDataset(args)
    ds.global_rank, ds.world_size, ds.real_epoch = 1, 2, 0
    assert len(ds) == args.epoch_steps * args.micro_bsz
    loader = data.make_loader(ds, args.micro_bsz, num_workers=workers)
    batches = list(data.DevicePrefetcher(loader, "cpu", dtype=torch.float32))
    assert len(batches) == args.epoch_steps
    idx, seen_images = 0, 0
    for b in batches:
        assert b["input_ids"].shape == (args.micro_bsz, args.ctx_len) and b["labels"].shape == b["input_ids"].shape
        k = 0                                                   # position in the batch's image stack
        for j in range(args.micro_bsz):
            i, rev = dp.rank_strided_sample(0, idx, 1, 2, args.epoch_steps * args.real_bsz, ds.magic_prime)
            rec = (list(reversed(recs)) if rev else recs)[i]
            ref = data.build_sample(rec, args.tokenizer, args.ctx_len, args.num_token_per_image)
            assert torch.equal(b["input_ids"][j], ref["input_ids"]) and torch.equal(b["labels"][j], ref["labels"])
            assert b["sample_id"][j] == str(rec["id"])
            if "image" in rec:
                im = torch.from_numpy(np.asarray(Image.open(os.path.join(args.image_folder, rec["image"])).convert("RGB")).copy())
                want = image.process_images([im], ("dino", "siglip", "sam"), torch.float32)
                for t, side in (("dino", 448), ("siglip", 448), ("sam", 1024)):
                    assert b["images"][t][k].shape == (3, side, side) and torch.equal(b["images"][t][k], want[t][0])
                k += 1
            idx += 1
        if k:
            assert b["images"]["num_image_per_sample"] == [1] * k and b["images"]["dino"].shape[0] == k
        seen_images += k
    assert seen_images > 0


def test_unreadable_image_becomes_zeros(tmp_path, gold):
    """dataset.py:213-215,236-241: a missing / unreadable image does not stop training; its towers are zero tensors."""
    from visualrwkv_amd import data
    recs, args = _make_dataset(tmp_path, gold, missing={r["image"] for r in gold["records"] if "image" in r})
    ds = data.MyDataset(args)
    with_images = [b for b in data.DevicePrefetcher(data.make_loader(ds, args.micro_bsz, num_workers=0), "cpu", dtype=torch.float32)
                   if "images" in b]
    assert with_images
    for b in with_images:
        n = sum(b["images"]["num_image_per_sample"])
        assert b["images"]["sam"].shape == (n, 3, 1024, 1024) and float(b["images"]["sam"].abs().sum()) == 0.0


def test_worker_side_transform_matches_reference_collate(tmp_path, gold):
    """decode_only=False: the reference's arrangement (transform inside the worker, multi_image_collate_fn)."""
    from visualrwkv_amd import data, image
    recs, args = _make_dataset(tmp_path, gold)

    def processor(pil):
        import numpy as np
        px = image.process_images([torch.from_numpy(np.asarray(pil).copy())], ("dino", "siglip", "sam"), torch.float32)
        return {k: v[0] for k, v in px.items()}

    recs = [r for r in recs if "image" in r]                   # the reference's collate needs an image in every batch
    import json
    (tmp_path / "data.json").write_text(json.dumps(recs))
    ds = data.MyDataset(args, decode_only=False, image_processor=processor)
    b = next(iter(data.make_loader(ds, args.micro_bsz, num_workers=0)))
    assert b["images"]["dino"].shape == (args.micro_bsz, 3, 448, 448) and b["images"]["num_image_per_sample"] == [1, 1]


@pytest.mark.gpu
def test_device_prefetcher_on_the_gpu(tmp_path, gold):
    """Same pipeline with the hand-off on a side stream and the tower transforms on the MI355X: equal (to float rounding of
    the resampler) to the CPU evaluation, delivered on the current stream."""
    from visualrwkv_amd import data
    recs, args = _make_dataset(tmp_path, gold)
    ds = data.MyDataset(args)
    cpu = list(data.DevicePrefetcher(data.make_loader(ds, args.micro_bsz, num_workers=0), "cpu", dtype=torch.float32))
    gpu = list(data.DevicePrefetcher(data.make_loader(ds, args.micro_bsz, num_workers=2), "cuda:0", dtype=torch.float32))
    assert len(cpu) == len(gpu) == args.epoch_steps
    for a, b in zip(cpu, gpu):
        assert b["input_ids"].is_cuda and torch.equal(a["input_ids"], b["input_ids"].cpu()) and torch.equal(a["labels"], b["labels"].cpu())
        assert ("images" in a) == ("images" in b)
        if "images" in a:
            assert a["images"]["num_image_per_sample"] == b["images"]["num_image_per_sample"]
            for t in ("dino", "siglip", "sam"):
                assert (a["images"][t] - b["images"][t].cpu()).abs().max() < 2e-3
