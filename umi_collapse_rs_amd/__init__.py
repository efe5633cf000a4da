"""MI355X-native drop-in for the UMI-collapse hot path of tkob-vh/umi-collapse-rs.

Product code: csrc/ (gfx950 HIP kernels + the C ABI of include/umihip.h) and this
thin host-side mirror of the reference's Algorithm / DataStruct interface.  There
is no CPU fallback: without libumihip.so and a gfx950 device every compute call
raises UmiHipError."""
from ._lib import (LIB_PATH, UMI_ALGO_ADJACENCY, UMI_ALGO_DIRECTIONAL, UMI_MAX_UMI_LEN, Stats,
                   UmiHipError, load)
from .api import Adjacency, Context, Directional, HipNaive, ReadFreq, default_context, to_bitset

__all__ = ["LIB_PATH", "UMI_ALGO_ADJACENCY", "UMI_ALGO_DIRECTIONAL", "UMI_MAX_UMI_LEN", "Stats",
           "UmiHipError", "load", "Adjacency", "Context", "Directional", "HipNaive", "ReadFreq",
           "default_context", "to_bitset"]
