"""arcquant_amd -- MI355X-native (gfx950) implementation of ARCQuant's NVFP4 + Augmented-Residual-Channel
GEMM hot path.

Layout of the package (only what the hot path needs):

* ``csrc/``     hand-written HIP kernels + the C-ABI (``include/arcq.h``) -> ``lib/libarcq_hip.so``
* ``_lib.py``   ctypes binding of that C-ABI (no torch types cross it)
* ``agemm.py``  drop-in mirror of the reference's pybind11 module ``agemm`` (kernels/src/bindings.cpp:551-575)
* ``qlinear.py`` host-side operator mirror: ``QLinearLayer``, ``NVFP4_reorder_quantize_{x,w}``
  (model/qLinearLayer.py, model/qLlamaLayer.py:73-77)
* ``tp.py``     tensor-parallel row / column sharding of a quantised linear over RCCL
"""
__version__ = "0.1.0"
