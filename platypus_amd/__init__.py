"""MI355X-native read-to-haplotype alignment and likelihood path of Platypus.

The work is done by hand-written HIP kernels in ``lib/libplatgpu.so`` behind the
C ABI of ``include/platgpu.h``; ``platypus_amd.native`` binds it.  Import of
``platypus_amd.native`` fails if the library is not built.
"""
__version__ = "0.1.0"
