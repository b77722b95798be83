"""Whole-region VCF text of the product against the CPU restatement of the reference's procedure
(oracle/vcf_oracle.py: variantcaller.pyx:116-191,251-439, cpopulation.pyx:189-513, vcf.py:710-739), from the
same buffered reads and candidates.  Bar: byte-identical body lines."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(seed=1),
    dict(seed=2, n_samples=3, coverage=20),
    dict(seed=3, paired=True, read_len=100, error_rate=0.01),
    dict(seed=4, spacing=25, n_var=120, indel_frac=0.4),          # crowded: multi-variant windows, multi-allelic sites
    dict(seed=5, coverage=12, error_rate=0.02),
    dict(seed=6, n_samples=2, half_covered=True),                 # an individual without reads: ./.:-1,-1,-1:-1:0
])
def test_region_vcf_lines_equal_the_restatement(kw):
    from platypus_amd import calign, pipeline
    from tests import vcf_compare as vc
    genome, truth, bufs = vc.simulated_region(**kw)
    got, want, skipped = vc.call_both(b"sim", 0, len(genome), bufs, genome, pipeline.Options(rlen=100))
    calign.closeDevice()
    diffs = vc.classify(got, want)
    assert len(want) > 10
    assert not diffs, {k: v[:3] for k, v in diffs.items()}
