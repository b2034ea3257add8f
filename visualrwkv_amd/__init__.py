"""MI355X-native hot path of VisualRWKV-7 (WKV7 operator, RWKV-7 blocks, ViT->projector feed).

Sub-modules import the gfx950 shared library lazily; `visualrwkv_amd.build.build()` compiles it.
"""
__version__ = "0.1.0"
