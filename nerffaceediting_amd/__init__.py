"""MI355X-native volumetric-rendering inference path of NeRFFaceEditing.

Host side: thin Python over the C-ABI library ``libnfe_render.so`` (hand-written HIP for gfx950).
There is no CPU or PyTorch fallback: every op raises if the library is missing or a tensor is not
on the GPU.
"""
__version__ = "0.1.0"
