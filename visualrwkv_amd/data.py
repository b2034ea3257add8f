"""Host-side batch construction with the reference's semantics (VisualRWKV-v7/v7.00/src/dataset.py:17-165):
conversation clean-up, "User:/Assistant:" templating with the "\\n\\n" round terminator, `<image>` -> a run of
`num_token_per_image` placeholder ids (65535), label masking of the human turns and of the 3-token "Assistant:"
prefix, truncate / pad to `ctx_len` (id 0, label -100) and the multi-image collate.  SURVEY.md 8(c) lists these as
harness behaviour the build must reproduce; they are pinned by tests/golden/data_ref.pt (the reference's own
`preprocess` on its dummy_data with its tokenizer).  The tokenizer itself is used unchanged: any object with
`encode(str) -> list[int]`.  Deterministic rank-strided sampling lives in dp.py (`rank_strided_sample`)."""
from __future__ import annotations

import copy
import re
from typing import Dict, List, Sequence

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = 65535
DEFAULT_IMAGE_TOKEN = "<image>"
STOP_TOKEN_INDEX = 261
DEFAULT_STOP_TOKEN = "\n\n"

_BLANK_LINES = re.compile(r"\n(\s*\n)+")
_ROLE = {"human": "User", "gpt": "Assistant"}
N_ASSISTANT_PREFIX_TOKENS = 3          # "Assistant:" tokenises to 3 ids with the RWKV world vocabulary (dataset.py:111-115)


def process_image_tokens_in_conversations(conversations: Sequence[Dict], num_image_paths: int) -> Sequence[Dict]:
    """Image placeholders first (one line each), then the text; blank-line runs collapse to one newline; the number of
    placeholders must equal the number of image files (dataset.py:39-67)."""
    total = sum(turn["value"].count(DEFAULT_IMAGE_TOKEN) for turn in conversations)
    assert total == num_image_paths, f"num_global_images: {total}, num_image_paths: {num_image_paths}, not match."
    for turn in conversations:
        text = turn["value"]
        if DEFAULT_IMAGE_TOKEN in text:
            n_here = text.count(DEFAULT_IMAGE_TOKEN)
            text = _BLANK_LINES.sub("\n", text.replace(DEFAULT_IMAGE_TOKEN, "").strip())
            if turn["from"].lower() == "human":
                text = "\n".join([DEFAULT_IMAGE_TOKEN] * n_here) + "\n" + text
            turn["value"] = text.strip()
        else:
            turn["value"] = _BLANK_LINES.sub("\n", text.strip())
    return conversations


def process_tokens_in_conversations(conversations: Sequence[Dict]) -> Sequence[Dict]:
    for turn in conversations:
        turn["value"] = _BLANK_LINES.sub("\n", turn["value"].strip())
    return conversations


def add_speaker_and_signal(conversations: Sequence[Dict]) -> Sequence[Dict]:
    """"User: ...\\n\\n" / "Assistant: ...\\n\\n"; an empty turn (inference prompt) becomes "Assistant:" (dataset.py:82-97)."""
    for turn in conversations:
        role = _ROLE.get(turn["from"].lower())
        if role is None:
            raise ValueError(f"Unknown speaker: {turn['from']}, must be human or gpt.")
        turn["value"] = f"{role}: {turn['value']}{DEFAULT_STOP_TOKEN}" if turn["value"] else f"{role}:"
    return conversations


def tokenize_with_image_token(prompt: str, tokenizer, num_token_per_image: int,
                              image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    pieces = [tokenizer.encode(chunk) for chunk in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids = list(pieces[0])
    for piece in pieces[1:]:
        ids += [image_token_index] * num_token_per_image
        ids += piece
    return ids


def mask_targets(targets: torch.Tensor, tokenized_lens: Sequence[int], speakers: Sequence[str]) -> None:
    pos = 0
    for n, who in zip(tokenized_lens, speakers):
        if who == "human":
            targets[pos:pos + n] = IGNORE_INDEX
        if who == "gpt":
            targets[pos:pos + N_ASSISTANT_PREFIX_TOKENS] = IGNORE_INDEX
        pos += n


def pad_to_max_len(input_ids: torch.Tensor, targets: torch.Tensor, max_len: int, pad_token_id: int):
    """Keep the first max_len tokens (the instruction stays complete), right-pad with id `pad_token_id` / label -100."""
    input_ids, targets = input_ids[:max_len], targets[:max_len]
    n = max_len - len(input_ids)
    if n > 0:
        input_ids = torch.cat([input_ids, torch.full((n,), pad_token_id, dtype=torch.long)])
        targets = torch.cat([targets, torch.full((n,), IGNORE_INDEX, dtype=torch.long)])
    return input_ids, targets


def preprocess(conversations, tokenizer, has_image: bool, ctx_len: int, num_token_per_image: int, pad_token_id: int = 0,
               do_pad_to_max_length: bool = True) -> Dict:
    """dataset.py:138-165.  Returns input_ids, labels (long tensors) and the concatenated input_text."""
    conversations = add_speaker_and_signal(conversations)
    text = "".join(turn["value"] for turn in conversations)
    ids: List[int] = []
    lens, speakers = [], []
    for turn in conversations:
        t = (tokenize_with_image_token(turn["value"], tokenizer, num_token_per_image) if has_image
             else tokenizer.encode(turn["value"]))
        ids += t
        lens.append(len(t))
        speakers.append(turn["from"])
    input_ids = torch.tensor(ids, dtype=torch.long)
    targets = input_ids.clone()
    mask_targets(targets, lens, speakers)
    if do_pad_to_max_length:
        input_ids, targets = pad_to_max_len(input_ids, targets, ctx_len, pad_token_id)
    return dict(input_ids=input_ids, labels=targets, input_text=text)


def build_sample(sample: Dict, tokenizer, ctx_len: int, num_token_per_image: int, pixel_values: Dict | None = None) -> Dict:
    """What MyDataset.__getitem__ returns for one JSON record (dataset.py:196-246), given the already processed images
    (`pixel_values`: tower name -> (n_images,3,H,W); None when the record has no image)."""
    if "image" in sample:
        n_img = 1 if isinstance(sample["image"], str) else len(sample["image"])
        conv = process_image_tokens_in_conversations(copy.deepcopy(sample["conversations"]), num_image_paths=n_img)
    else:
        conv = process_tokens_in_conversations(copy.deepcopy(sample["conversations"]))
    out = preprocess(conv, tokenizer, has_image="image" in sample, ctx_len=ctx_len, num_token_per_image=num_token_per_image)
    if "image" in sample:
        out["images"] = pixel_values if pixel_values is not None else {
            "dino": torch.zeros(n_img, 3, 448, 448), "siglip": torch.zeros(n_img, 3, 448, 448),
            "sam": torch.zeros(n_img, 3, 1024, 1024)}
    out["sample_id"] = sample["sample_id"] if "sample_id" in sample else sample["id"]
    return out


def multi_image_collate_fn(batch: Sequence[Dict]) -> Dict:
    """dataset.py:23-36: stack ids/labels, concatenate every tower's images over the samples that have any."""
    with_img = [x for x in batch if "images" in x]
    images = {k: torch.cat([x["images"][k] for x in with_img], dim=0) for k in ("dino", "sam", "siglip")}
    images["num_image_per_sample"] = [len(x["images"]["dino"]) for x in with_img]
    return dict(input_text=[x["input_text"] for x in batch], input_ids=torch.stack([x["input_ids"] for x in batch]),
                labels=torch.stack([x["labels"] for x in batch]), images=images, sample_id=[str(x["sample_id"]) for x in batch])
