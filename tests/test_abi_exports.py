"""The C-ABI library loads on a machine without a GPU and exports every
function include/platgpu.h declares (no compute is called here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "platgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from platypus_amd import native
    names = _declared()
    assert len(names) >= 14
    lib = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(native.SYMBOLS) == names
    assert native.LIB.pg_abi_version() == native.ABI_VERSION


def test_no_device_is_an_error_not_a_fallback():
    import torch  # only to ask whether this machine has a GPU
    from platypus_amd import native
    if torch.cuda.is_available():
        return
    try:
        native.AlignContext()
    except native.PlatGpuError as e:
        assert e.code == native.PG_ERR_NO_DEVICE
    else:
        raise AssertionError("context creation must fail without a device")


def test_nibble_packing_is_the_bam_code():
    """pack_nibbles writes what a BAM record holds (bam1_seq: two bases per byte, the first in the high half, codes of
    bam_nt16_rev_table "=ACMGRSVTWYHKDBN", platypus/c/samtools/bam.h)"""
    import numpy as np
    from platypus_amd import native
    assert native.NIBBLE_CODES == b"=ACMGRSVTWYHKDBN"
    seq = np.frombuffer(b"ACGTN=M", dtype=np.uint8)
    assert native.pack_nibbles(seq).tolist() == [0x12, 0x48, 0xF0, 0x30]
    assert native.pack_nibbles(seq[:0]).tolist() == []
