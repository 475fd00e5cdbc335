"""MI355X-native nonlinear-refinement back end (bundle adjustment, nonlinear triangulation,
nonlinear PnP) for the willSapgreen/structure-from-motion pipeline.

The directory name carries a hyphen, so import it with
``importlib.import_module("structure-from-motion_amd")`` or through the root-level alias
module ``sfm_amd`` (``import sfm_amd``).

Layout
------
csrc/          hand-written HIP kernels (gfx950) + the C-ABI (include/sfm_hip.h) -> libsfm_hip.so
native.py      ctypes binding of the C-ABI; raises if the library is missing (no CPU fallback)
processors.py  drop-in mirrors of the reference's *_processor classes for the hot path
observations.py  KeyTrack tables -> observation CSR with the reference's is_visible semantics
sampling.py    RANSAC subsets from Python's RNG stream, drawn in bulk with random.sample's exact consumption
sharding.py    point-range sharding of BA across ranks (one process per GPU, RCCL all-reduce)
geometry.py    host-side q<->R helpers (camera block packing, reference exceptions)
scenes.py      seeded synthetic scenes of BASELINE.json's configs
"""
from . import geometry, native, observations, processors, sampling, scenes, sharding  # noqa: F401

__all__ = ["geometry", "native", "observations", "processors", "sampling", "scenes", "sharding"]
