"""MI355X-native VideoTokenizer hot path (encode -> quantize -> decode) behind the
magvit2_pytorch.VideoTokenizer API.  Kernels: libmv2_hip.so (include/mv2.h)."""
from .tokenizer import VideoTokenizer, MagViT2, __version__
from .discriminator import Discriminator
from .spec import README_LAYERS

__all__ = ['VideoTokenizer', 'MagViT2', 'Discriminator', 'README_LAYERS', '__version__']
