"""CPU checks of the restatement of platypus/cpopulation.pyx (setup, EM, variant posterior).

The reference ships no fixture for this stage and its Cython layer cannot be imported here, so the
restatement is cross-checked against an independent numpy formulation (matrix form of the same
sums) and its two loop orders are compared with each other."""
import numpy as np
import pytest

from oracle import pyoracle as po


def numpy_caller(like, off, max_iters=100):
    """independent formulation: full symmetric matrices, y = P f"""
    H, R = like.shape
    n_ind = len(off) - 1
    iu = np.triu_indices(H)
    ll = np.zeros((n_ind, len(iu[0])))
    has = np.array([off[i + 1] > off[i] for i in range(n_ind)])
    for i in range(n_ind):
        if has[i]:
            seg = like[:, off[i]:off[i + 1]]
            ll[i] = np.log(0.5 * (seg[iu[0]] + seg[iu[1]])).sum(axis=1)
    mx = max(-1e7, ll[has].max())
    post = np.exp(ll - mx)
    P = np.zeros((n_ind, H, H))
    P[:, iu[0], iu[1]] = post
    P[:, iu[1], iu[0]] = post
    f = np.full(H, 1.0 / H)
    eps = min(1e-3, 1.0 / (n_ind * 4))
    change, iters = eps + 1, 0
    while change > eps and iters < max_iters:
        tot = np.zeros(H)
        for i in np.flatnonzero(has):
            y = P[i] @ f
            tot += 2 * f * y / (f @ y)
        nf = tot / (2 * has.sum())
        change = np.abs(nf - f).max()
        f, iters = nf, iters + 1
    return post, f, iters


@pytest.mark.parametrize("H,R,n_ind,seed", [(2, 10, 1, 1), (4, 32, 1, 2), (5, 40, 3, 3), (12, 90, 6, 4), (30, 120, 9, 5)])
def test_restatement_matches_matrix_form(H, R, n_ind, seed):
    rng = np.random.default_rng(seed)
    like = np.minimum(1.0, np.exp(-rng.random((H, R)) * rng.choice([2.0, 30.0], size=(H, 1))))
    cuts = np.sort(rng.integers(0, R + 1, size=n_ind - 1))
    off = np.concatenate([[0], cuts, [R]]).astype(np.int32)
    post = po.caller_setup(like, off)
    a = po.caller_call(post, off, H, literal=True, want_cisr=True)
    b = po.caller_call(post, off, H, literal=False, want_cisr=True)
    for k in ("freq", "best_gen", "best_val", "cisr"):
        assert np.array_equal(a[k], b[k]), k                      # same additions in the same order
    assert a["iters"] == b["iters"] and a["max_change"] == b["max_change"]
    n_post, n_f, n_it = numpy_caller(like, off)
    np.testing.assert_allclose(post, n_post, rtol=1e-10)
    np.testing.assert_allclose(a["freq"], n_f, rtol=1e-9, atol=1e-300)
    assert a["iters"] == n_it
    np.testing.assert_allclose(a["freq"].sum(), 1.0, rtol=1e-12)


def test_known_small_case():
    # two haplotypes, one individual, every read fits haplotype 1 ten times better
    like = np.array([[0.01] * 8, [0.1] * 8])
    off = np.array([0, 8], dtype=np.int32)
    post = po.caller_setup(like, off)
    ll = np.array([8 * np.log(0.01), 8 * np.log(0.5 * 0.11), 8 * np.log(0.1)])
    np.testing.assert_allclose(post[0], np.exp(ll - ll.max()), rtol=1e-13)
    res = po.caller_call(post, off, 2)
    assert res["best_gen"][0] == 2 and res["freq"][1] > 0.99          # homozygous for haplotype 1
    # a variant carried by haplotype 1 only: removing it leaves hom-ref, whose likelihood is 1e-8 of the best
    p = po.variant_posterior(post, off, res["freq"], [0, 1], 1e-3)
    assert p > 0.999
    # a "variant" no haplotype carries changes nothing: posterior == prior
    np.testing.assert_allclose(po.variant_posterior(post, off, res["freq"], [0, 0], 0.25), 0.25, rtol=1e-12)


def test_individual_without_reads():
    rng = np.random.default_rng(9)
    like = np.exp(-rng.random((3, 20)) * 10)
    off = np.array([0, 12, 12, 20], dtype=np.int32)
    post = po.caller_setup(like, off)
    res = po.caller_call(post, off, 3)
    assert res["best_gen"][1] == -1 and res["best_val"][1] == 0.0     # cpopulation.pyx:777-780
    # its normalised row is exp(0 - max), cpopulation.pyx:717-718,734
    assert np.all(post[1] == post[1][0]) and post[1][0] >= 1.0
