"""farkle_ii_amd — MI355X-native engine for the Farkle_II simulation hot path.

Python host code mirrors the reference's operator surface for this path (strategies, RNG
coordinates, ``_play_one_shuffle``/``_run_chunk`` tallies, ``simulate_many_games``, the H2H
``BlockRunner``) and calls a C-ABI shared library (``include/farkle_hip.h``) of hand-written HIP
kernels for gfx950.  There is no CPU fallback: simulation entry points raise if the HIP library
or a GPU is missing.
"""
__version__ = "0.1.0"
