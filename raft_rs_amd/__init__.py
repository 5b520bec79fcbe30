"""raft_rs_amd -- MI355X-native multi-raft progress/commit engine.

One hot path of pingcap/raft-rs (leader-side MsgAppendResponse handling + commit-index
recompute, see include/raftgroups.h) for millions of raft groups per call, as hand-written
HIP kernels behind a C ABI (raft_rs_amd/libraftgroups.so). This package is the thin Python
binding used by tests and bench.py; there is no CPU fallback: without the built HIP library
importing `raft_rs_amd.engine` raises.
"""
from .engine import (Engine, EngineError, MsgBuffers, load_library, LIB_PATH,  # noqa: F401
                     COL, MF, PF, OUT, cfg_make, WL_MAJORITY, WL_JOINT, WL_MIXED,
                     VARIANT_DEFAULT, VARIANT_LANE, VARIANT_LDS, VARIANT_COOP, VARIANT_LDS_DMA, VARIANT_COMPACT,
                     CACHE, CFGF, KERNEL)

__all__ = ["Engine", "EngineError", "MsgBuffers", "load_library", "LIB_PATH", "COL", "MF", "PF", "OUT",
           "cfg_make", "WL_MAJORITY", "WL_JOINT", "WL_MIXED", "VARIANT_DEFAULT", "VARIANT_LANE",
           "VARIANT_LDS", "VARIANT_COOP", "VARIANT_LDS_DMA", "VARIANT_COMPACT", "CACHE", "CFGF", "KERNEL"]
