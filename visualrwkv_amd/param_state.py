"""Process-wide parameter generation counter.

`Zero1Engine.step()` updates parameters through raw pointers (the fused AdamW kernel writes the bf16 shard, RCCL gathers
into the flat buffer), which neither changes `Tensor._version` nor `data_ptr()`.  Everything that caches something derived
from parameter VALUES (transposed LoRA factors and captured HIP graphs of the decode path) includes `generation()` in its
cache key; every optimizer step bumps it.
"""
_GENERATION = 0


def generation() -> int:
    return _GENERATION


def bump() -> int:
    global _GENERATION
    _GENERATION += 1
    return _GENERATION
