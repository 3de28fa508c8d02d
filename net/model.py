"""Drop-in module path of the reference (`from net.model import PromptIR`, train.py:10,
test.py:14, demo.py:8).  The implementation is the MI355X-native one in promptir_amd.model."""
from promptir_amd.model import (  # noqa: F401
    Attention, BiasFree_LayerNorm, Downsample, FeedForward, LayerNorm, OverlapPatchEmbed, PromptGenBlock,
    PromptIR, TransformerBlock, Upsample, WithBias_LayerNorm,
)
