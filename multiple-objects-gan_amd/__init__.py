"""mogan_amd: MI355X-native (gfx950) GAN training hot path of tohinz/multiple-objects-gan.

Loaded under the alias `mogan_amd` by /mogan_loader.py. Sub-packages:
  csrc/     hand-written HIP kernels + the C ABI (libmogan_hip.so, include/mogan_hip.h)
  hip/      ctypes binding + torch.autograd.Function wrappers (the only callers of the ABI)
  attngan/  host-side mirror of code/coco/attngan/{model,GlobalAttention,trainer}.py,
            miscc/{config,losses,utils}.py of the reference
"""
__version__ = "0.1.0"
