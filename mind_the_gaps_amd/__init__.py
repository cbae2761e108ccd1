"""mind_the_gaps_amd -- MI355X-native engine for mind_the_gaps' celerite GP
log-likelihood hot path (GPModelling._log_probability / fit / derive_posteriors).

Python host code mirroring the reference API on top of a C-ABI
(include/mtg.h, libmtg_hip.so) into hand-written gfx950 HIP kernels.
"""
__version__ = "0.1.0"
