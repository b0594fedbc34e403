"""viquae_amd -- MI355X (gfx950) implementation of ViQuAE's dense-retrieval hot path.

Scope: ``meerqat.ir.search`` (brute-force kNN behind ``KnowledgeBase``),
``meerqat.ir.embedding`` / ``meerqat.image.embedding`` (DPR/BERT and CLIP-ViT encoders).  The
arithmetic runs in hand-written HIP kernels (``csrc/``) reached through the C ABI declared in
``include/meerqat_hip.h``; this package is the Python host side that mirrors the reference's
call surface.  There is no CPU fallback: every compute entry point raises if the HIP library
or a GPU is missing.
"""
__version__ = "0.1.0"
