"""ecoz2rs_amd -- MI355X-native VQ hot path of mbari-org/ecoz2rs (`vq learn` / `vq quantize`).

The compute lives in hand-written HIP kernels inside ``csrc/libecoz2vq.so`` (C-ABI declared in
``include/ecoz2_vq.h``).  This package is the thin host-side mirror of the reference's
``ecoz2_lib`` wrappers (``/root/reference/src/ecoz2_lib/mod.rs:252-342``) plus the session
API used by ``bench.py`` and the tests.  There is no CPU fallback: importing works without a
GPU (so symbols can be inspected), every compute call raises when no HIP device is usable.
"""
from ._lib import lib, lib_path, Ecoz2Error, check  # noqa: F401
from .vq import (  # noqa: F401
    VqGroup,
    VqSession,
    LevelStats,
    vq_learn,
    vq_quantize,
    vq_classify,
    vq_show,
    version,
)
from . import classify, formats, hmm, synth  # noqa: F401
