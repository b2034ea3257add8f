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


# ---------------------------------------------------------------------------------------------------------------
# Dataset, loader and device prefetcher (VisualRWKV-v7/v7.00/src/dataset.py:167-246, train.py:219-222)
#
# The reference decodes AND transforms every image (three bicubic resizes, one to 1024 x 1024) in its single DataLoader
# worker.  Here the workers only decode (PIL -> uint8 HWC); the three tower transforms run on the GPU
# (visualrwkv_amd.image.process_images) on a side stream, one batch ahead of the training step, together with the
# host-to-device copies from pinned memory.  Sampling, templating, masking and collation are the reference's.
# ---------------------------------------------------------------------------------------------------------------
class MyDataset(torch.utils.data.Dataset):
    """dataset.py:167-246.  `args` needs: data_file, image_folder, tokenizer, ctx_len, num_token_per_image, epoch_steps,
    real_bsz, micro_bsz (the reference's names).  `global_rank`, `world_size` and `real_epoch` are attributes the trainer
    sets, as in the reference (train_callback).  decode_only=True (default) returns the decoded uint8 images under
    'images_u8' for the device transform; decode_only=False applies `image_processor` (a callable image -> dict of
    (3,S,S) tensors, as args.image_processor of the reference) in the worker."""

    def __init__(self, args, decode_only: bool = True, image_processor=None):
        import json
        from .dp import largest_3n_plus_2_prime
        self.args = args
        self.tokenizer = args.tokenizer
        with open(args.data_file, "r") as f:
            self.list_data_dict = json.load(f)
        self.list_data_dict_reverse = list(reversed(self.list_data_dict))
        self.data_size = len(self.list_data_dict)
        self.magic_prime = largest_3n_plus_2_prime(self.data_size)
        self.samples_per_epoch = args.epoch_steps * args.real_bsz
        self.decode_only = decode_only
        self.image_processor = image_processor
        self.global_rank, self.world_size, self.real_epoch = 0, 1, 0

    def __len__(self):
        return self.args.epoch_steps * self.args.micro_bsz

    def record(self, idx: int) -> Dict:
        from .dp import rank_strided_sample
        i, rev = rank_strided_sample(self.real_epoch, idx, self.global_rank, self.world_size, self.samples_per_epoch, self.magic_prime)
        return (self.list_data_dict_reverse if rev else self.list_data_dict)[i]

    def __getitem__(self, idx):
        import os
        sample = copy.deepcopy(self.record(idx))
        images_u8, pixel_values = None, None
        if "image" in sample:
            names = [sample["image"]] if isinstance(sample["image"], str) else list(sample["image"])
            sample["image"] = names
            try:
                from PIL import Image
                import numpy as np
                decoded = [Image.open(os.path.join(self.args.image_folder, n)).convert("RGB") for n in names]
                if self.decode_only:
                    images_u8 = [torch.from_numpy(np.asarray(im).copy()) for im in decoded]           # (H,W,3) uint8
                else:
                    per = [self.image_processor(im) for im in decoded]
                    pixel_values = {k: torch.stack([p[k] for p in per], dim=0) for k in per[0]}
            except Exception:                                    # unreadable image: zero tensors, as the reference (dataset.py:213-215)
                images_u8, pixel_values = None, None
        out = build_sample(sample, self.tokenizer, self.args.ctx_len, self.args.num_token_per_image, pixel_values)
        if "image" in sample and self.decode_only:
            if images_u8 is not None:
                del out["images"]
                out["images_u8"] = images_u8
            else:
                out["images_missing"] = len(sample["image"])
        return out


def decode_collate_fn(batch: Sequence[Dict]) -> Dict:
    """Collation for decode-only samples: ids / labels stacked as in multi_image_collate_fn, decoded images kept as a flat
    list of uint8 (H,W,3) tensors (sizes differ) with the per-sample counts; samples whose images could not be read carry
    their count in 'missing' (zero tensors after the transform, as the reference)."""
    imgs, counts = [], []
    for x in batch:
        if "images_u8" in x:
            imgs.extend(x["images_u8"]); counts.append(len(x["images_u8"]))
        elif "images" in x or "images_missing" in x:
            n = x.get("images_missing", len(x["images"]["dino"]) if "images" in x else 0)
            imgs.extend([None] * n); counts.append(n)
    return dict(input_text=[x["input_text"] for x in batch], input_ids=torch.stack([x["input_ids"] for x in batch]),
                labels=torch.stack([x["labels"] for x in batch]), images_u8=imgs, num_image_per_sample=counts,
                sample_id=[str(x["sample_id"]) for x in batch])


def make_loader(dataset: "MyDataset", micro_bsz: int, num_workers: int = 4):
    """train.py:219-222 with decode-only workers: shuffle=False (the dataset does the rank-strided sampling), drop_last,
    pinned memory; more than the reference's single worker is safe because __getitem__ depends on idx only."""
    return torch.utils.data.DataLoader(dataset, collate_fn=decode_collate_fn if dataset.decode_only else multi_image_collate_fn,
                                       shuffle=False, pin_memory=torch.cuda.is_available(), batch_size=micro_bsz,
                                       num_workers=num_workers, persistent_workers=False, drop_last=True)


class DevicePrefetcher:
    """Iterates a loader one batch ahead: host-to-device copies and the three tower transforms of batch i+1 are issued on a
    side stream while the training step of batch i runs; `next()` makes the current stream wait for them.  Yields the
    `samples` dict VisualRWKV.forward takes (input_ids, labels, images{dino,siglip,sam,num_image_per_sample}, sample_id)."""

    def __init__(self, loader, device, towers=("dino", "siglip", "sam"), dtype=torch.bfloat16):
        self.loader, self.device, self.towers, self.dtype = loader, torch.device(device), tuple(towers), dtype
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _to_device(self, batch):
        from .image import TOWER_SPECS, process_images
        nb = self.device.type == "cuda"
        out = {"input_ids": batch["input_ids"].to(self.device, non_blocking=nb), "labels": batch["labels"].to(self.device, non_blocking=nb),
               "sample_id": batch["sample_id"], "input_text": batch.get("input_text")}
        if "images_u8" in batch:
            if batch["images_u8"]:
                per_tower = {t: [] for t in self.towers}
                for im in batch["images_u8"]:
                    if im is None:
                        for t in self.towers:
                            per_tower[t].append(torch.zeros(1, 3, TOWER_SPECS[t][0], TOWER_SPECS[t][0], dtype=self.dtype, device=self.device))
                    else:
                        im = im.pin_memory() if nb and not im.is_pinned() else im
                        px = process_images([im.to(self.device, non_blocking=nb)], self.towers, self.dtype)
                        for t in self.towers:
                            per_tower[t].append(px[t])
                images = {t: torch.cat(v, dim=0) for t, v in per_tower.items()}
                images["num_image_per_sample"] = batch["num_image_per_sample"]
                out["images"] = images
        elif "images" in batch:
            out["images"] = {k: (v.to(self.device, dtype=self.dtype, non_blocking=nb) if torch.is_tensor(v) else v)
                             for k, v in batch["images"].items()}
        return out

    def __iter__(self):
        it = iter(self.loader)

        def produce():
            try:
                host = next(it)
            except StopIteration:
                return None
            if self.stream is None:
                return self._to_device(host), None
            with torch.cuda.stream(self.stream):
                dev = self._to_device(host)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return dev, ev

        nxt = produce()
        while nxt is not None:
            cur, ev = nxt
            nxt = produce()                                   # batch i+1 is in flight while the caller trains on batch i
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in list(cur.values()) + list(cur.get("images", {}).values()):
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield cur
