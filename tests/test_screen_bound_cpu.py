"""CPU: the error bound that makes the fp16 screening pass exact (csrc/scan_screen.hip header), checked
numerically on a numpy emulation of the screen's arithmetic -- fp16(64 x) images, exact fp16 x fp16 products, fp32
accumulation -- against the fp32 k-ordered fmaf chain the exact kernel and the re-score use.

    |s~ - s_fp32| <= |dx|max |q| + |x|max |dq| + |dx|max |dq| + 5e-5 |x|max |q|

and the consequence the kernel relies on: when the sufficiency test passes, the exact top-k is inside the screen's top-K'."""
import numpy as np
import pytest


def image(v):
    return (v.astype(np.float32) * np.float32(64.0)).astype(np.float16)


def screen_scores(x, q):
    hx, hq = image(x).astype(np.float32), image(q).astype(np.float32)      # products of two fp16 values are exact in fp32
    acc = np.zeros((q.shape[0], x.shape[0]), np.float32)
    for k0 in range(0, x.shape[1], 16):                                   # one MFMA step = 16 k, accumulated in fp32
        acc += (hq[:, k0:k0 + 16] @ hx[:, k0:k0 + 16].T).astype(np.float32)
    return acc * np.float32(1.0 / 4096.0)


def exact_scores_fp32(x, q):
    acc = np.zeros((q.shape[0], x.shape[0]), np.float32)
    for k in range(x.shape[1]):                                           # k-ordered chain of fp32 multiply-adds
        acc = (acc.astype(np.float64) + q[:, k:k + 1].astype(np.float64) * x[None, :, k].astype(np.float64)).astype(np.float32)
    return acc


def eps(x, q):
    xd, qd = x.astype(np.float64), q.astype(np.float64)
    dx = np.linalg.norm(xd - image(x).astype(np.float64) / 64.0, axis=1).max()
    dq = np.linalg.norm(qd - image(q).astype(np.float64) / 64.0, axis=1)
    xn, qn = np.linalg.norm(xd, axis=1).max(), np.linalg.norm(qd, axis=1)
    return dx * qn + xn * dq + dx * dq + 5e-5 * xn * qn


def corpora():
    rng = np.random.default_rng(3)
    unit = rng.standard_normal((3000, 384)).astype(np.float32)
    unit /= np.linalg.norm(unit, axis=1, keepdims=True)
    yield "unit", unit, unit[:40] + 0.1 * rng.standard_normal((40, 384)).astype(np.float32)
    big = (rng.standard_normal((2000, 384)) * rng.uniform(0.05, 20.0, (2000, 1))).astype(np.float32)
    yield "norms 0.05..400", big, (rng.standard_normal((30, 384)) * 7).astype(np.float32)
    sparse = np.zeros((2000, 384), np.float32)
    for i in range(2000):                                                  # few large components + fp16-subnormal dust
        sparse[i, rng.integers(0, 384, 6)] = rng.standard_normal(6)
        sparse[i, rng.integers(0, 384, 40)] += rng.standard_normal(40).astype(np.float32) * 1e-7
    yield "sparse + subnormal dust", sparse, sparse[:30] + 1e-3 * rng.standard_normal((30, 384)).astype(np.float32)
    one_sided = np.abs(unit[:1500]) + 0.5                                  # all-positive: errors cannot cancel
    yield "all positive", one_sided, np.abs(unit[:20]) + 0.25


@pytest.mark.parametrize("name,x,q", list(corpora()), ids=[c[0] for c in corpora()])
def test_screen_error_bound_holds(name, x, q):
    err = np.abs(screen_scores(x, q).astype(np.float64) - exact_scores_fp32(x, q).astype(np.float64))
    bound = eps(x, q)[:, None]
    assert (err <= bound).all(), (name, float((err / bound).max()))
    assert (err / bound).max() > 1e-3                                       # the bound is not vacuous on these inputs


def test_sufficiency_test_implies_containment():
    """Whenever s~[K'-1] < s~[k-1] - 2 EPS, the exact top-k (fp32 scores, ties by lower row) lies inside the screen's top-K'."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((6000, 384)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[100:140] = x[100] + 2e-4 * rng.standard_normal((40, 384)).astype(np.float32)      # a tight cluster: some queries must fail
    q = np.concatenate([x[rng.permutation(6000)[:60]] + 0.2 * rng.standard_normal((60, 384)).astype(np.float32), x[100:104]])
    st, se, e = screen_scores(x, q), exact_scores_fp32(x, q), eps(x, q)
    k, kp, passed, failed = 10, 32, 0, 0
    for i in range(q.shape[0]):
        order_t = np.lexsort((np.arange(6000), -st[i]))[:kp]
        if st[i, order_t[kp - 1]] < st[i, order_t[k - 1]] - 2 * e[i]:
            passed += 1
            exact = np.lexsort((np.arange(6000), -se[i]))[:k]
            assert set(exact) <= set(order_t), i
        else:
            failed += 1
    assert passed >= 50 and failed >= 1            # both branches exercised (the cluster queries fail the test)


# ---- the native L2 index (round 5): the row norm starts the screening chain as its C operand -------------------------------------------------
def l2_screen_half_scores(x, q):
    """scan_screen_lean3_kernel<.., L2N = 1>: acc starts at -2048 |x|^2 (fp32, the stored norm scaled exactly), then the 24 MFMA steps; the
    candidate score is acc / 4096 = q~.x~ - |x|^2 / 2."""
    n2 = (x.astype(np.float32) ** 2).sum(1, dtype=np.float32)              # the stored column (k_l2_aug_rows; any fp32 summation order)
    hx, hq = image(x).astype(np.float32), image(q).astype(np.float32)
    acc = np.broadcast_to((np.float32(-2048.0) * n2)[None, :], (q.shape[0], x.shape[0])).astype(np.float32).copy()
    for k0 in range(0, x.shape[1], 16):
        acc += (hq[:, k0:k0 + 16] @ hx[:, k0:k0 + 16].T).astype(np.float32)
    return acc * np.float32(1.0 / 4096.0), n2


def l2_exact_half_scores_fp32(x, q, n2):
    """The exact L2 scan's chain: fp32 multiply-adds over (2q, x) in k order, then fma(-|x|^2, 1, .) -- halved (exact) for comparison."""
    s2 = exact_scores_fp32(x, (2.0 * q).astype(np.float32))
    full = (s2.astype(np.float64) - n2[None, :].astype(np.float64)).astype(np.float32)
    return full * np.float32(0.5)


def eps_l2(x, q):
    xn = np.linalg.norm(x.astype(np.float64), axis=1).max()
    return eps(x, q) + 1.5e-5 * xn * xn                                     # k_rescore<true> (scan_screen.hip)


@pytest.mark.parametrize("name,x,q", [c for c in corpora() if c[0] != "norms 0.05..400"] + [
    ("norms 0.05..30", (np.random.default_rng(5).standard_normal((2000, 384)) * np.random.default_rng(6).uniform(0.0025, 1.5, (2000, 1))).astype(np.float32),
     (np.random.default_rng(7).standard_normal((30, 384)) * 0.7).astype(np.float32))], ids=lambda v: v if isinstance(v, str) else None)
def test_l2_screen_error_bound_holds(name, x, q):
    """|half score of the screen - half score of the exact L2 chain| <= EPS(q) + 1.5e-5 |x|max^2: the image errors bound the q.x part as for the
    inner product; the norm is the same stored number on both sides and only the roundings that see it are new."""
    ap, n2 = l2_screen_half_scores(x, q)
    ex = l2_exact_half_scores_fp32(x, q, n2)
    err = np.abs(ap.astype(np.float64) - ex.astype(np.float64))
    bound = eps_l2(x, q)[:, None]
    assert (err <= bound).all(), (name, float((err / bound).max()))
    # and the ranking it protects is the L2 ranking: the exact half score orders rows like -|q - x|^2 up to fp32 rounding
    d2 = ((q[:, None, :].astype(np.float64) - x[None, :200, :].astype(np.float64)) ** 2).sum(-1)
    half = (q.astype(np.float64) ** 2).sum(1)[:, None] - 2.0 * ex[:, :200].astype(np.float64)
    assert np.allclose(half, d2, rtol=0, atol=2e-4 * max(1.0, float(d2.max())))
