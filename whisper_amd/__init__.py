"""whisper_amd: MI355X (gfx950) implementation of ucbsky/whisper's server hot path.

The compute lives in libwhisper_gfx950.so (hand-written HIP kernels + C ABI, include/whisper_gfx950.h);
this package is the thin host-side mirror of the reference's Rust call sites used by the tests and the
benchmark.  There is no CPU fallback: importing fails if the library is missing, and every compute
call fails if no GPU is present.
"""
from ._lib import lib, LIB_PATH, WhError  # noqa: F401
from .prio3 import (  # noqa: F401
    Prio3, Prio3Params, new_sum_vec_256, new_histogram_256, new_average_256, shard_batched_many,
    client_set_test_sampler_bound, flp_encode, flp_prove, SUMVEC, HISTOGRAM, AVERAGE,
)
from .grouptest import general_binary_split_test, ClientProofTag  # noqa: F401
from . import hh  # noqa: F401,E402
from .count import Prio3Count, new_count  # noqa: F401,E402

SORT_NOALLOC_MAX_N = 1 << 20   # WH_SORT_NOALLOC_MAX_N of include/whisper_gfx950.h
