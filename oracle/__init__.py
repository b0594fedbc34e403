"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``viquae_amd/`` does (the product path fails loudly when the HIP
library is missing instead of falling back to anything here).

* :mod:`oracle.knn`      -- ctypes front-end of ``knn_oracle.c`` (brute-force IP/L2 top-k, L2norm,
  shard merge) + a slow numpy cross-check used to validate the C file itself.
* :mod:`oracle.encoders` -- numpy fp32 restatement of the BERT/DPR and CLIP-ViT forward passes.

Parity status is stated in each module's header and in DESIGN.md.
"""
