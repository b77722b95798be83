"""Whole-region VCF text of the product against the CPU restatement of the reference's procedure
(oracle/vcf_oracle.py: variantcaller.pyx:116-191,251-439, cpopulation.pyx:189-513, vcf.py:710-739), from the
same buffered reads and candidates.  Bar: byte-identical body lines."""
import numpy as np
import pytest

from tests import vcf_compare as vc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(seed=1),
    dict(seed=2, n_samples=3, coverage=20),
    dict(seed=3, paired=True, read_len=100, error_rate=0.01),
    dict(seed=4, spacing=25, n_var=120, indel_frac=0.4),          # crowded: multi-variant windows, multi-allelic sites
    dict(seed=5, coverage=12, error_rate=0.02),
    dict(seed=6, n_samples=2, half_covered=True),                 # an individual without reads: ./.:-1,-1,-1:-1:0
    dict(seed=11, n_samples=50, coverage=10, G=8000, n_var=30),   # joint calling of 50 samples at 10x
    dict(seed=12, n_samples=24, coverage=10, G=8000, n_var=60, spacing=40, indel_frac=0.4),     # ... with crowded windows
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


def _three_sample_window(first_single):
    """three samples at different coverage around one window: a six-base deletion (first sample, when first_single), and
    inside its span a SNP (second sample) and an insertion (third sample) at ONE position"""
    from platypus_amd import simulate
    rng = np.random.default_rng(41)
    G = 6000
    genome = simulate.random_genome(rng, G)
    p = 3000
    while len(set(genome[p - 2:p + 12])) < 4 or genome[p + 4:p + 5] == b"A":   # no repeat to shift the indels in
        p += 1
    dele = (p, genome[p + 1:p + 7], b"", 2)
    snp = (p + 4, genome[p + 4:p + 5], b"A", 2)
    ins = (p + 4, b"", b"GTC" if genome[p + 5:p + 6] != b"G" else b"CTG", 2)
    bufs = []
    for mine, cov in (([dele] if first_single else [], 18), ([snp], 30), ([ins], 44)):
        reads = simulate.simulate_reads(rng, genome, mine, coverage=cov, read_len=100, error_rate=0.0)
        bufs.append(simulate.buffer_of(reads, 0, G))
    return genome, bufs, p


def test_read_count_column_at_a_position_with_several_variants():
    """variantcaller.pyx:422-423,436: the NR column is a local that only a position with one variant assigns.  At a
    position with two variants every called individual shows what the last such position left there (the last
    individual's count); without one before it the reference fails there and the rest of the window is never written
    (:189-191).  Product and restatement do the same."""
    from platypus_amd import calign, pipeline
    opts = pipeline.Options(rlen=100, minPosterior=0)
    genome, bufs, p = _three_sample_window(first_single=True)
    got, want, skipped = vc.call_both(b"sim", 0, len(genome), bufs, genome, opts)
    assert not skipped and got == want
    rows = [l.split("\t") for l in got]
    assert len(rows) == 2
    single, double = rows
    assert "," not in single[4] and "," in double[4] and int(double[1]) == p + 5
    own = [c.split(":")[3] for c in single[9:]]
    assert len(set(own)) == 3                                            # three individuals, three read counts
    assert [c.split(":")[3] for c in double[9:]] == [own[-1]] * 3        # ... and the last one's in every column after it
    # nothing with one variant before the position with two: the window is lost from there on
    genome, bufs, p = _three_sample_window(first_single=False)
    got, want, skipped = vc.call_both(b"sim", 0, len(genome), bufs, genome, opts)
    calign.closeDevice()
    assert len(skipped) == 1 and "nReadsThisIndividual" in str(skipped[0][2])
    assert got == want == []


@pytest.mark.gpu
@pytest.mark.parametrize("inflate", ["device", "host"])
def test_file_to_file_run_equals_the_restatements_end_to_end(tmp_path, inflate):
    """BAM + FASTA files in, VCF file out through platypus_amd.runner (BGZF blocks inflated on the GPU or by the host
    library, device kernels), against the CPU restatements chained from the records on: buffer filtering and trimming and
    candidates (oracle/candidates_oracle.py), windows to VCF text (oracle/vcf_oracle.py).  Byte-identical body lines."""
    got, want, skipped = vc.files_both(str(tmp_path), 31, inflate=inflate)
    assert not skipped
    assert len(want) > 80
    diff = vc.classify(got, want)
    assert not diff, {k: v[:3] for k, v in diff.items()}
    assert got == want


@pytest.mark.gpu
def test_file_to_file_run_of_fifty_bam_files_equals_the_restatements(tmp_path):
    """the joint-calling configuration from files: 50 BAM files (one read buffer each per stretch,
    variantcaller.pyx:82-90; one individual each, cpopulation.pyx:698-725) -> one VCF with 50 sample columns, against
    the chained CPU restatements.  Byte-identical body lines.  (Each stretch's 50 files are read side by side by the
    runner's helper threads.)"""
    got, want, skipped = vc.files_both(str(tmp_path), 77, inflate="device", G=30000, unit=10000, chunk=15000, n_var=36,
                                       spacing=350, coverage=7, n_samples=50, homopolymers=8)
    assert not skipped
    assert len(want) > 20 and all(len(l.split("\t")) == 9 + 50 for l in want)
    diff = vc.classify(got, want)
    assert not diff, {k: v[:3] for k, v in diff.items()}
    assert got == want


def test_equal_posteriors_stay_equal_in_the_ranking_of_a_crowded_window(tmp_path):
    """a window with more candidates than maxVariants keeps the ones with the highest posteriors (variantcaller.pyx:142-166);
    here four SNPs with two supporting reads each stand beside a deletion with exactly equal inputs, their posteriors are
    equal to the last bit in the reference's arithmetic, and the tie goes by position.  The device adds the posterior's
    sums in the reference's order for such windows, so the same three variants survive (a run tests/fuzz_files.py found:
    hardened records, two samples, maxVariants 3)."""
    kw = dict(inflate="device", G=80794, unit=5000, rlen=100, n_var=373, spacing=40, indel_frac=0.6, coverage=30, paired=True,
              error_rate=0.02, n_samples=2, chunk=5000, homopolymers=0, options=(("minPosterior", 20), ("maxVariants", 3)),
              harden=True)
    got, want, skipped = vc.files_both(str(tmp_path), 582, **kw)
    assert len(want) > 300
    diff = vc.classify(got, want)
    assert not diff, {k: v[:3] for k, v in diff.items()}
    assert got == want
