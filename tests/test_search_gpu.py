"""GPU parity tests of the dense search path, through the C-ABI (librmu.so), against the CPU oracle.

Bar (north_star): identical top-k row ids (bit-exact integer result; a swap is tolerated only between
entries whose fp64 oracle scores differ by < 1e-6), scores within 1e-4.
At BASELINE.json's full sizes the oracle cannot run in seconds, so size-independent properties are checked.
"""
import os
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_topk_parity

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rmu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import ragmeup_amd
    from ragmeup_amd import _native
    _native.lib()                       # fails loudly if librmu.so is missing: there is no fallback
    return ragmeup_amd


@pytest.fixture(scope="module")
def corpus50k():
    x = O.make_corpus(50_000)
    q, planted = O.make_queries(x, 256)
    return x, q, planted


# geometry coverage: WQ=1 (nq<=32), WQ=2 (<=64), WQ=4; k classes <=32 and <=112; ragged last tiles
@pytest.mark.parametrize("nq,k", [(1, 10), (7, 10), (32, 20), (33, 10), (64, 32), (65, 10), (200, 10),
                                  (256, 1), (130, 100), (40, 112), (5, 33)])
def test_parity_vs_oracle(rmu, corpus50k, nq, k):
    x, q, planted = corpus50k
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(q[:nq], k)
    os_, or_ = O.flat_search(q[:nq], x, k)
    assert_topk_parity(s, r, os_, or_)
    assert (r[:, 0] == planted[:nq]).all()
    assert (np.diff(s, axis=1) <= 0).all()                       # best first
    idx.close()


@pytest.mark.parametrize("n", [1, 7, 31, 32, 33, 127, 128, 129, 4097])
def test_tiny_and_ragged_corpora(rmu, n):
    x = O.make_corpus(n)
    q, _ = O.make_queries(x, 5)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    for k in (1, 10, 40):
        s, r = idx.search(q, k)
        os_, or_ = O.flat_search(q, x, k)
        assert_topk_parity(s, r, os_, or_)
        if n < k:                                                # fewer live rows than k: (-inf, -1) padding
            assert (r[:, n:] == -1).all() and np.isneginf(s[:, n:]).all()
    idx.close()


@pytest.mark.parametrize("n", [1, 7, 31, 33, 129, 4097])
@pytest.mark.parametrize("nq", [130, 300])
def test_tiny_corpora_on_the_screening_path(rmu, n, nq):
    """Full query tiles over a handful of rows: one partial tile, fewer rows than K' = 32, padding beyond the live rows."""
    x = O.make_corpus(n, seed=41)
    q = O.make_corpus(nq, seed=42)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    for k in (1, 10, 24):
        s, r = idx.search(q, k)
        assert idx.last_screened() != 0
        os_, or_ = O.flat_search(q, x, min(k + 4, max(n, 1)))
        assert_topk_parity(s[:, :min(k, n)], r[:, :min(k, n)], os_, or_)
        if n < k:
            assert (r[:, n:] == -1).all() and np.isneginf(s[:, n:]).all()
    idx.close()


def test_empty_index(rmu):
    idx = rmu.FlatIndex(384)
    s, r = idx.search(np.ones((3, 384), np.float32), 10)
    assert (r == -1).all() and np.isneginf(s).all()
    idx.close()


def test_committed_golden_fixture(rmu):
    g = np.load(os.path.join(HERE, "golden", "search_golden.npz"))
    x = O.make_corpus(int(g["n"]), int(g["d"]), int(g["seed_x"]))
    q, _ = O.make_queries(x, int(g["nq"]), int(g["seed_q"]))
    idx = rmu.FlatIndex(int(g["d"]))
    idx.add(x)
    s, r = idx.search(q, int(g["k"]))
    assert_topk_parity(s, r, g["scores"], g["rows"])
    idx.close()


def test_duplicates_resolve_to_lower_row(rmu):
    x = O.make_corpus(3000)
    xd = np.concatenate([x, x[100:110], x[100:110]])            # each of rows 100..109 exists three times
    idx = rmu.FlatIndex(384)
    idx.add(xd)
    s, r = idx.search(x[100:110], 3)
    for i in range(10):
        assert list(r[i]) == [100 + i, 3000 + i, 3010 + i]       # exact ties: ascending row id
    os_, or_ = O.flat_search(x[100:110], xd, 3)
    assert np.array_equal(r, or_)
    idx.close()


def test_incremental_add_growth_and_row_ids(rmu):
    x = O.make_corpus(30_000)
    q, _ = O.make_queries(x, 16)
    idx = rmu.FlatIndex(384, capacity_hint=16)                   # forces several reallocations
    firsts = [idx.add(x[lo:lo + 1000]) for lo in range(0, 30_000, 1000)]   # RAGHelper._batch_size = 1000
    assert firsts == list(range(0, 30_000, 1000)) and len(idx) == 30_000
    s, r = idx.search(q, 10)
    assert_topk_parity(s, r, *O.flat_search(q, x, 10))
    got = idx.get_rows([0, 999, 1000, 29_999])
    assert np.array_equal(got, x[[0, 999, 1000, 29_999]])       # MMR re-fetch is bit-exact
    idx.close()


def test_tombstones(rmu):
    x = O.make_corpus(20_000)
    q, planted = O.make_queries(x, 64)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    dead = np.unique(np.concatenate([planted[:32], np.arange(5000, 5100)]))
    assert idx.remove_rows(dead) == dead.size
    assert idx.remove_rows(dead[:5]) == 0                        # already gone
    alive = np.ones(20_000, bool); alive[dead] = False
    s, r = idx.search(q, 10)
    assert_topk_parity(s, r, *O.flat_search(q, x, 10, alive=alive))
    assert not np.isin(r, dead).any()
    idx.close()


@pytest.mark.parametrize("d", [64, 192, 256, 384, 512, 768])
def test_other_dimensions(rmu, d):
    x = O.make_corpus(6000, d, seed=5)
    q, _ = O.make_queries(x, 40, seed=6)
    idx = rmu.FlatIndex(d)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert_topk_parity(s, r, *O.flat_search(q, x, 10))
    assert np.array_equal(idx.get_rows([17]), x[[17]])
    idx.close()


def test_cosine_metric_on_unnormalised_rows(rmu):
    from ragmeup_amd import _native as N
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((8000, 384)) * rng.uniform(0.2, 5.0, (8000, 1))).astype(np.float32)
    q = (rng.standard_normal((20, 384)) * 3.0).astype(np.float32)
    idx = rmu.FlatIndex(384, metric=N.METRIC_COSINE)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert_topk_parity(s, r, *O.flat_search(q, x, 10, O.METRIC_COSINE))
    idx.close()


def test_cosine_metric_on_the_screening_path(rmu):
    """COSINE index (what the vector store creates): rows stored normalised, queries normalised per call, full query tiles."""
    from ragmeup_amd import _native as N
    rng = np.random.default_rng(19)
    x = (rng.standard_normal((40_000, 384)) * rng.uniform(0.2, 5.0, (40_000, 1))).astype(np.float32)
    q = (x[rng.permutation(40_000)[:300]] + 0.3 * rng.standard_normal((300, 384))).astype(np.float32) * 2.5
    idx = rmu.FlatIndex(384, metric=N.METRIC_COSINE)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert idx.last_screened() != 0
    assert_topk_parity(s, r, *O.flat_search(q, x, 14, O.METRIC_COSINE))
    s2, r2 = idx.search(q, 112)                      # (k > 104: the exact fp32 scan)
    assert idx.last_screened() == 0 and np.array_equal(r2[:, :10], r) and np.array_equal(s2[:, :10], s)
    idx.close()


def test_device_pointers_and_row_base(rmu, corpus50k):
    import torch
    x, q, _ = corpus50k
    idx = rmu.FlatIndex(384)
    idx.add(torch.from_numpy(x).cuda())                          # device -> device append
    s, r = idx.search(torch.from_numpy(q[:100]).cuda(), 10, row_base=1_000_000)
    assert s.is_cuda and r.dtype == torch.int64
    os_, or_ = O.flat_search(q[:100], x, 10)
    assert_topk_parity(s.cpu().numpy(), r.cpu().numpy(), os_, or_ + 1_000_000)
    idx.close()


def test_topk_merge_abi(rmu):
    x = O.make_corpus(12_000)
    q, _ = O.make_queries(x, 70)
    ps, pr = [], []
    for lo in range(0, 12_000, 1500):                            # 8 shards, as on an 8-GPU node
        s, r = O.flat_search(q, x[lo:lo + 1500], 10)
        ps.append(s.astype(np.float32)); pr.append(r + lo)
    ms, mr = rmu.topk_merge(np.stack(ps), np.stack(pr))
    gs, gr = O.flat_search(q, x, 10)
    assert_topk_parity(ms, mr, gs, gr)
    # short lists (-1 padded) and k > 64
    s, r = O.flat_search(q[:3], x[:5], 100)
    ms, mr = rmu.topk_merge(np.stack([s.astype(np.float32)] * 2), np.stack([r, np.where(r >= 0, r + 5, -1)]))
    assert (mr[:, 10:] == -1).all() and (np.sort(mr[:, :10], axis=1) == np.arange(10)).all()


def test_emulated_shards_equal_global(rmu, corpus50k):
    """8 row shards on one GPU + rmu_topk_merge == one global search (the N>1 data path minus the all-gather)."""
    from ragmeup_amd.shard import shard_bounds
    x, q, _ = corpus50k
    parts_s, parts_r = [], []
    for rank in range(8):
        lo, hi = shard_bounds(x.shape[0], 8, rank)
        idx = rmu.FlatIndex(384)
        idx.add(x[lo:hi])
        s, r = idx.search(q, 10, row_base=lo)
        parts_s.append(s); parts_r.append(r)
        idx.close()
    ms, mr = rmu.topk_merge(np.stack(parts_s), np.stack(parts_r))
    assert_topk_parity(ms, mr, *O.flat_search(q, x, 10))


def test_concurrent_searches_and_writer(rmu, corpus50k):
    """Flask's threaded server + LCEL RunnableParallel call search from several threads while /add_document
    may append: searches are re-entrant, add takes the writer lock."""
    x, q, _ = corpus50k
    idx = rmu.FlatIndex(384)
    idx.add(x[:40_000])
    want_s, want_r = O.flat_search(q[:48], x[:40_000], 10)
    errs = []

    def reader(i):
        try:
            for _ in range(5):
                s, r = idx.search(q[i * 8:(i + 1) * 8], 10)
                # rows >= 40000 may legitimately appear once the writer has appended them
                m = r < 40_000
                if m.all():
                    assert_topk_parity(s, r, want_s[i * 8:(i + 1) * 8], want_r[i * 8:(i + 1) * 8])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def writer():
        try:
            idx.add(x[40_000:])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=reader, args=(i,)) for i in range(6)] + [threading.Thread(target=writer)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    s, r = idx.search(q[:48], 10)
    assert_topk_parity(s, r, *O.flat_search(q[:48], x, 10))
    idx.close()


def test_invalid_requests_raise(rmu):
    from ragmeup_amd._native import RmuError
    idx = rmu.FlatIndex(384)
    idx.add(O.make_corpus(100))
    with pytest.raises(RmuError):
        idx.search(np.ones((1, 384), np.float32), 0)
    with pytest.raises(RmuError):
        idx.search(np.ones((1, 384), np.float32), 113)
    with pytest.raises(ValueError):
        idx.search(np.ones((1, 100), np.float32), 10)
    with pytest.raises(RmuError):
        idx.get_rows([100])
    with pytest.raises(RmuError):
        rmu.FlatIndex(4096)
    idx.close()


def test_full_size_properties_1m(rmu):
    """BASELINE config 2 (1M x 384, B=1024, top-10): properties that need no O(N*B) oracle pass, plus an
    oracle check of a 32-query subset."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((1_000_000, 384), generator=g, device="cuda")
    x /= x.norm(dim=1, keepdim=True)
    pick = torch.randperm(1_000_000, generator=g, device="cuda")[:1024]
    q = x[pick] + 0.1 * torch.randn((1024, 384), generator=g, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    idx = rmu.FlatIndex(384, capacity_hint=1_000_000)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert (r[:, 0] == pick).all()                                         # planted neighbour found
    assert (s[:, :-1] >= s[:, 1:]).all()                                   # sorted
    assert (r.sort(dim=1).values.diff(dim=1) > 0).all()                    # ids unique per query
    # re-score the returned ids independently (fp64 gather-dot) and check nothing better was missed:
    rescored = torch.einsum("qkd,qd->qk", x[r].double(), q.double())
    assert (rescored - s.double()).abs().max() < 1e-4
    full = (q[:64].double() @ x.double().T)                                # 64 x 1M fp64 on the device (checker only)
    kth = torch.topk(full, 10, dim=1).values[:, -1]
    assert (s[:64, -1].double() - kth).abs().max() < 1e-6
    # oracle parity on a subset
    xs = x.cpu().numpy(); qs = q[:32].cpu().numpy()
    assert_topk_parity(s[:32].cpu().numpy(), r[:32].cpu().numpy(), *O.flat_search(qs, xs, 10))
    # idempotence
    s2, r2 = idx.search(q, 10)
    assert torch.equal(r, r2) and torch.equal(s, s2)
    # (round 6) full batches below 6M rows take the ladder ratio with the fewest levels: 1M rows -> 1 952 / 15 616 / 124 992 / 1M (ratio 8)
    # instead of ratio 3's five; whatever the geometry, the answers are the exact scan's
    assert idx.last_screened() != 0 and idx.last_geometry()["launches"] == 4
    for ratio, first, launches in ((3, 256, 5), (8, 2048, 3)):
        idx.set_ladder(ratio, first)
        s3, r3 = idx.search(q, 10)
        assert idx.last_geometry()["launches"] == launches and torch.equal(r, r3) and torch.equal(s, s3)
    idx.set_ladder(0, 0)
    idx.close()


def test_rccl_allgather_path_world1(rmu, corpus50k):
    """The collective leg (pack -> RCCL all_gather_into_tensor -> unpack -> rmu_topk_merge) at world_size 1 --
    the only size a 1-GPU box allows (SURVEY.md 8e iii); world_size 2 is covered on CPU with gloo."""
    import os
    import torch
    import torch.distributed as dist
    from ragmeup_amd.shard import ShardedSearcher
    x, q, _ = corpus50k
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        idx = rmu.FlatIndex(384)
        idx.add(x)
        ss = ShardedSearcher(idx, row_base=0, force_collective=True)
        s, r = ss.search(torch.from_numpy(q[:96]).cuda(), 10)
        assert_topk_parity(s.cpu().numpy(), r.cpu().numpy(), *O.flat_search(q[:96], x, 10))
        idx.close()
    finally:
        if created:
            dist.destroy_process_group()


# ---- fp16 screening pass + exact fp32 re-score (k <= 32, dim 384; RMU_OPT_SCREEN_MIN_NQ = 1 where the batch is small) ----------------------
@pytest.mark.parametrize("nq,k", [(1, 20), (7, 10), (64, 24), (128, 10), (200, 1), (256, 16), (1024, 10), (7, 25), (200, 32), (1024, 28)])
def test_screened_search_matches_oracle_and_exact_path(rmu, nq, k):
    """(round 5: 24 < k <= 32 screens with K' = 40 kept candidates)"""
    x = O.make_corpus(60_000)
    q, planted = O.make_queries(x, nq)
    idx = rmu.FlatIndex(384)
    idx.set_screen_min_batch(1)                      # small batches over a small corpus would take the exact scan (it is faster there)
    idx.add(x)
    s, r = idx.search(q, k)
    assert idx.last_screened() != 0, "expected the screening path to answer this batch"   # (< 0: some queries were re-run exactly)
    assert_topk_parity(s, r, *O.flat_search(q, x, k + 4))      # oracle a few ranks deeper: boundary near-ties
    assert (r[:, 0] == planted).all()
    # bit-identical to the exact fp32 scan (same summation order in the re-score)
    s2, r2 = idx.search(q, 112)                      # k > 104 always takes the exact scan
    assert idx.last_screened() == 0
    assert np.array_equal(r2[:, :k], r) and np.array_equal(s2[:, :k], s)
    idx.close()


@pytest.mark.parametrize("n,nq,k", [(300_000, 384, 10), (262_144 + 17, 129, 16), (700_001, 1024, 5), (300_000, 300, 32), (262_144 + 17, 129, 27)])
def test_screened_ladder_matches_oracle_and_exact_path(rmu, n, nq, k):
    """n >= 262144 rows: the corpus is scanned as a ladder of row ranges whose merged K'-th best seeds the next launch's
    thresholds (ragged last tiles, partial query tiles, 1- and 2-group wave geometries)."""
    x = O.make_corpus(n, seed=21)
    q, planted = O.make_queries(x, nq, seed=22)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(q, k)
    assert idx.last_screened() != 0, "expected the screening path"
    assert idx.last_geometry()["launches"] >= 3, "expected a multi-launch ladder"
    s2, r2 = idx.search(q, 112)                      # exact fp32 scan (k > 104)
    assert idx.last_screened() == 0
    assert np.array_equal(r2[:, :k], r) and np.array_equal(s2[:, :k], s)          # bit-identical
    assert (r[:, 0] == planted).all()
    sub = slice(0, 96)                               # oracle (fp64 numpy) on a subset of the queries: seconds, not minutes
    assert_topk_parity(s[sub], r[sub], *O.flat_search(q[sub], x, k + 4))
    idx.close()


def test_screened_ladder_after_deletes_and_reload(rmu, tmp_path):
    x = O.make_corpus(280_000, seed=23)
    q, planted = O.make_queries(x, 200, seed=24)
    idx = rmu.FlatIndex(384)
    idx.add(x[:150_000]); idx.add(x[150_000:])
    dead = np.unique(planted[:60])
    idx.remove_rows(dead)
    s, r = idx.search(q, 10)
    assert idx.last_screened() != 0 and not np.isin(r, dead).any()
    s2, r2 = idx.search(q, 112)                      # (k > 104: the exact fp32 scan)
    assert idx.last_screened() == 0 and np.array_equal(r2[:, :10], r) and np.array_equal(s2[:, :10], s)
    path = str(tmp_path / "big.rmu")
    idx.save(path)
    idx2 = rmu.FlatIndex.load(path)
    s3, r3 = idx2.search(q, 10)
    assert idx2.last_screened() != 0
    assert np.array_equal(r3, r) and np.array_equal(s3, s)
    idx.close(); idx2.close()


def test_screened_search_falls_back_on_dense_ties(rmu):
    """More than K' = 32 exact duplicates of the best row: the sufficiency test must flag that query and the exact scan
    must answer it (only it: the other 199 stay on the screened path) -- ids in ascending-row order among the ties."""
    x = O.make_corpus(20_000)
    xd = np.concatenate([x, np.repeat(x[77:78], 40, axis=0)])            # row 77 exists 41 times
    q = np.concatenate([x[77:78], O.make_queries(x, 199)[0]])
    idx = rmu.FlatIndex(384)
    idx.add(xd)
    s, r = idx.search(q, 10)
    assert idx.last_screened() == -1                                          # exactly one query was re-run
    assert_topk_parity(s, r, *O.flat_search(q, xd, 10))
    assert list(r[0]) == [77] + list(range(20_000, 20_009))
    # every query hits the duplicates: more than 1/8 of the batch fails -> the whole batch goes to the exact scan
    q2 = np.repeat(x[77:78], 128, axis=0) + np.float32(1e-4) * O.make_queries(x, 128)[0]
    s2, r2 = idx.search(q2, 10)
    assert idx.last_screened() <= -100
    assert_topk_parity(s2, r2, *O.flat_search(q2, xd, 14))
    idx.close()


def test_screened_search_with_tombstones_and_growth(rmu):
    x = O.make_corpus(40_000)
    q, planted = O.make_queries(x, 160)
    idx = rmu.FlatIndex(384, capacity_hint=64)
    for lo in range(0, 40_000, 8000):
        idx.add(x[lo:lo + 8000])
    dead = np.unique(planted[:50])
    idx.remove_rows(dead)
    alive = np.ones(40_000, bool); alive[dead] = False
    s, r = idx.search(q, 10)
    assert idx.last_screened() > 0
    assert_topk_parity(s, r, *O.flat_search(q, x, 10, alive=alive))
    assert not np.isin(r, dead).any()
    idx.close()


def test_screened_search_unnormalised_rows(rmu):
    """Un-normalised rows (norms 0.1 .. 60) on the screening path: ids identical to the fp64 oracle under the tie rule, and
    bit-identical to the exact fp32 scan.  Scores are O(50) here: the north-star 1e-4 / 1e-6 bars are stated for unit-norm
    scores, so both are scaled by the score magnitude."""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((30_000, 384)) * rng.uniform(0.1, 3.0, (30_000, 1))).astype(np.float32)
    q = rng.standard_normal((130, 384)).astype(np.float32)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert idx.last_screened() != 0, "130 queries must take the screening path"
    os_, or_ = O.flat_search(q, x, 14)
    scale = max(1.0, float(np.abs(os_).max()))
    assert_topk_parity(s, r, os_, or_, score_tol=1e-4 * scale, tie_tol=1e-6 * scale)
    idx.set_screening(False)
    s2, r2 = idx.search(q, 10)
    assert idx.last_screened() == 0
    assert np.array_equal(r2, r) and np.array_equal(s2, s)
    idx.close()


def test_save_load_roundtrip(rmu, tmp_path):
    x = O.make_corpus(25_000)
    q, _ = O.make_queries(x, 140)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    idx.remove_rows([5, 6, 7])
    s0, r0 = idx.search(q, 10)
    path = str(tmp_path / "corpus.rmu")
    idx.save(path)
    idx.close()
    assert os.path.getsize(path) == 64 + 25_000 + 25_000 * 1536
    idx2 = rmu.FlatIndex.load(path)
    assert len(idx2) == 25_000
    s1, r1 = idx2.search(q, 10)                        # screened path on the reloaded (re-split) image
    assert np.array_equal(r0, r1) and np.array_equal(s0, s1)
    s2, r2 = idx2.search(q[:5], 10)                    # exact path
    assert np.array_equal(r2, r0[:5])
    assert not np.isin(r1, [5, 6, 7]).any()            # tombstones survive
    assert idx2.add(x[:10]) == 25_000                  # and the index keeps growing
    idx2.close()
    with pytest.raises(Exception):
        rmu.FlatIndex.load(str(tmp_path / "missing.rmu"))


# ---- batched device-side MMR (rmu_index_mmr) -------------------------------------------------------------------
@pytest.mark.parametrize("lam", [0.5, 0.0, 1.0, 0.3])
def test_device_mmr_matches_oracle(rmu, lam):
    x = O.make_corpus(20_000, seed=31)
    q, _ = O.make_queries(x, 96, seed=32, noise=0.4)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(q, 20)
    pos = idx.mmr(q, r, 10, lam)
    assert pos.shape == (96, 10) and pos.dtype == np.int32
    for i in range(96):
        want = O.mmr(q[i], x[r[i]], k=10, lambda_mult=lam)
        assert list(pos[i]) == want, (i, list(pos[i]), want)
    idx.close()


def test_device_mmr_edges(rmu):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((7, 384)).astype(np.float32)
    x[5] = x[2]                                        # duplicate candidate: identical scores, lowest position wins
    idx = rmu.FlatIndex(384)
    idx.add(x)
    q = rng.standard_normal((3, 384)).astype(np.float32)
    s, r = idx.search(q, 20)                           # only 7 live rows: padded with -1
    assert (r[:, 7:] == -1).all()
    pos = idx.mmr(q, r, 10)
    for i in range(3):
        want = O.mmr(q[i], x[r[i, :7]], k=10)
        assert list(pos[i, :7]) == want and (pos[i, 7:] == -1).all()
    # k = 1 is the nearest neighbour; a query equal to a stored row picks that row first
    assert (idx.mmr(q, r, 1)[:, 0] == 0).all()
    pos = idx.mmr(x[3:4], idx.search(x[3:4], 5)[1], 3)
    assert pos[0, 0] == 0
    from ragmeup_amd._native import RmuError
    with pytest.raises(RmuError):
        idx.mmr(q, np.zeros((3, 65), np.int64), 4)      # fetch_k > 64
    idx.close()


@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
def test_search_mmr_in_one_call_equals_the_two_calls(rmu, metric):
    """rmu_index_search_mmr == rmu_index_search + rmu_index_mmr (rows in pick order, their search scores), incl. a row_base, fewer
    live rows than fetch_k, and the error cases."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((5000, 384)).astype(np.float32)
    q = (x[:33] + 0.6 * rng.standard_normal((33, 384))).astype(np.float32)
    from ragmeup_amd import _native as NN
    metric = {"ip": NN.METRIC_IP, "cosine": NN.METRIC_COSINE, "l2": NN.METRIC_L2SQ}[metric]
    idx = rmu.FlatIndex(384, metric=metric)
    idx.add(x)
    s, r = idx.search(q, 20)
    pos = idx.mmr(q, r, 10, 0.35)
    rows, sc = idx.search_mmr(q, 20, 10, 0.35, row_base=1000)
    assert (rows == np.take_along_axis(r, pos.astype(np.int64), axis=1) + 1000).all()
    assert (sc == np.take_along_axis(s, pos.astype(np.int64), axis=1)).all()
    one_r, _ = idx.search_mmr(q[5], 20, 10, 0.35)                                  # the per-request shape: one query
    assert (one_r[0] == np.take_along_axis(r, pos.astype(np.int64), axis=1)[5]).all()
    idx.close()
    tiny = rmu.FlatIndex(384, metric=metric)
    tiny.add(x[:6])
    rows, sc = tiny.search_mmr(q[:2], 20, 10)
    assert ((rows[:, :6] >= 0).all() and (rows[:, 6:] == -1).all() and np.isneginf(sc[:, 6:]).all()
            and sorted(rows[0, :6].tolist()) == list(range(6)))
    from ragmeup_amd._native import RmuError
    with pytest.raises(RmuError):
        tiny.search_mmr(q[:2], 65, 4)
    with pytest.raises(RmuError):
        tiny.search_mmr(q[:2], 8, 9)
    tiny.close()


@pytest.mark.parametrize("dim", [100, 768])
def test_device_mmr_other_dims(rmu, dim):
    """dim <= 384 takes the LDS-staged kernel (one workgroup per query), wider rows the one-wave-per-query kernel: same picks."""
    rng = np.random.default_rng(dim)
    x = rng.standard_normal((3000, dim)).astype(np.float32)
    q = (x[:40] + 0.5 * rng.standard_normal((40, dim))).astype(np.float32)
    idx = rmu.FlatIndex(dim)
    idx.add(x)
    s, r = idx.search(q, 32)
    pos = idx.mmr(q, r, 12, 0.4)
    for i in range(40):
        assert list(pos[i]) == O.mmr(q[i], x[r[i]], k=12, lambda_mult=0.4)
    idx.close()


def test_vectorstore_batch_mmr_uses_device_and_matches_single(rmu):
    from ragmeup_amd.vectorstore import MI355XVectorStore
    from ragmeup_amd.documents import Document

    class Emb:                                           # deterministic toy embedding: hash -> unit vector
        def _v(self, t):
            g = np.random.default_rng(abs(hash(t)) % (2 ** 32))
            v = g.standard_normal(384).astype(np.float32)
            return v / np.linalg.norm(v)
        def embed_documents(self, texts):
            return [self._v(t).tolist() for t in texts]
        def embed_query(self, t):
            return self._v(t).tolist()

    docs = [Document(page_content=f"chunk {i}", metadata={"source": "s", "id": str(i)}) for i in range(500)]
    st = MI355XVectorStore.from_documents(docs, Emb(), collection_name="mmr")
    qs = [f"question {i}" for i in range(9)]
    batch = st.max_marginal_relevance_search_batch(qs, k=5, fetch_k=20)
    for qtext, got in zip(qs, batch):
        single = st.max_marginal_relevance_search(qtext, k=5, fetch_k=20)    # the reference's per-request call: device selection too
        assert [d.page_content for d in got] == [d.page_content for d in single]
        # ... and both equal the host fp64 expression (langchain's) on the re-fetched candidate vectors
        from ragmeup_amd.vectorstore import maximal_marginal_relevance
        qv = np.asarray(Emb().embed_query(qtext), dtype=np.float32)
        _, r = st._index.search(qv[None], 20)
        want = [st._doc(int(r[0][i])).page_content for i in maximal_marginal_relevance(qv, st._index.get_rows(r[0]), 5, 0.5)]
        assert [d.page_content for d in got] == want


# ---- round 2: parity pinned at the headline configuration and on the hardware ------------------------------------------------
def _fp64_topk_on_device(x, q, k, chunk=500_000):
    """Independent checker: chunked fp64 scores (torch matmul on the device) + running top-k by (score desc, row asc)."""
    import torch
    best_s = torch.full((q.shape[0], k), -float("inf"), dtype=torch.float64, device=q.device)
    best_r = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=q.device)
    qd = q.double()
    for lo in range(0, x.shape[0], chunk):
        hi = min(x.shape[0], lo + chunk)
        sc = qd @ x[lo:hi].double().T
        cs, ci = torch.topk(sc, min(k, hi - lo), dim=1)          # ties inside a chunk: resolved below by the stable sort on rows
        alls = torch.cat([best_s, cs], dim=1)
        allr = torch.cat([best_r, ci + lo], dim=1)
        # order by (score desc, row asc): sort by row first, then a stable sort by -score
        o1 = torch.argsort(torch.where(allr < 0, torch.full_like(allr, 2 ** 62), allr), dim=1, stable=True)
        alls, allr = torch.gather(alls, 1, o1), torch.gather(allr, 1, o1)
        o2 = torch.argsort(-alls, dim=1, stable=True)[:, :k]
        best_s, best_r = torch.gather(alls, 1, o2), torch.gather(allr, 1, o2)
        del sc
    return best_s, best_r


def test_headline_10m_parity_vs_independent_fp64(rmu):
    """BASELINE headline configuration -- 10M x 384 fp32 rows, ONE 1024-query batch, top-10, default path (the 7-launch
    screening ladder + fp32 re-score) -- against an independent fp64 computation of the whole batch on the device:
    ids identical under the tie rule, scores within 1e-4 (north_star), row ids above 2^23 included."""
    import torch
    N, B, K = 10_000_000, 1024, 10
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.empty((N, 384), dtype=torch.float32, device="cuda")
    for lo in range(0, N, 1 << 20):
        hi = min(N, lo + (1 << 20))
        t = torch.randn((hi - lo, 384), generator=g, dtype=torch.float32, device="cuda")
        x[lo:hi] = t / t.norm(dim=1, keepdim=True)
    pick = torch.randperm(N, generator=g, device="cuda")[:B]
    q = x[pick] + 0.1 * torch.randn((B, 384), generator=g, dtype=torch.float32, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    idx = rmu.FlatIndex(384, capacity_hint=N)
    idx.add(x)
    s, r = idx.search(q, K)
    assert idx.last_screened() != 0 and idx.last_geometry()["launches"] >= 5, "expected the screening ladder"
    ref_s, ref_r = _fp64_topk_on_device(x, q, K + 4)
    assert (r[:, 0] == pick).all() and (pick > (1 << 23)).any()
    assert (s.double() - ref_s[:, :K]).abs().max().item() <= 1e-4
    bad = (r != ref_r[:, :K]).nonzero()
    for qi, pi in bad.tolist():                                   # a swap is legal only between fp64 near-ties
        where = (ref_r[qi] == r[qi, pi]).nonzero()
        assert where.numel() == 1, f"query {qi} pos {pi}: row {int(r[qi, pi])} is not in the fp64 top-{K + 4}"
        assert abs(float(ref_s[qi, where[0, 0]] - ref_s[qi, pi])) <= 1e-6
    assert bad.shape[0] <= B * K // 1000                          # and such near-ties are rare on this corpus
    # the exact fp32 scan (API switch) returns the same bits
    idx.set_screening(False)
    s2, r2 = idx.search(q[:128], K)
    assert idx.last_screened() == 0
    assert torch.equal(r2, r[:128]) and torch.equal(s2, s[:128])
    idx.close()


def test_screen_error_bound_on_hardware(rmu):
    """The containment argument rests on |s~ - s_fp32| <= EPS(q).  The CPU test checks it on a numpy emulation; this one
    checks the KERNEL: the screening pass's candidate scores (v_mfma_f32_32x32x16_f16 accumulation on the MI355X) against
    the exact fp32 scores of the same rows and against fp64, for the same four corpora."""
    from tests.test_screen_bound_cpu import corpora, eps as eps_numpy
    for name, x, q in corpora():
        idx = rmu.FlatIndex(384)
        idx.add(x)
        ap, rows, ex, eps = idx.screen_candidates(q)
        valid = rows >= 0
        assert valid.sum() >= q.shape[0] * 16, name
        err32 = np.abs(ap.astype(np.float64) - ex.astype(np.float64))
        assert (err32[valid] <= np.broadcast_to(eps[:, None], ap.shape)[valid]).all(), (name, float((err32 / eps[:, None])[valid].max()))
        true = np.einsum("qcd,qd->qc", x[np.where(valid, rows, 0)].astype(np.float64), q.astype(np.float64))
        err64 = np.abs(ap.astype(np.float64) - true)
        assert (err64[valid] <= np.broadcast_to(eps[:, None], ap.shape)[valid]).all(), name
        # the device's EPS is the documented formula (measured |dx|max, per-query |dq|): never below the numpy value
        assert (eps >= 0.999 * eps_numpy(x, q)).all() and (eps <= 1.05 * eps_numpy(x, q) + 1e-12).all(), name
        # exact scores of the candidates are what an independent fp64 dot gives (fp32 accumulation error only)
        assert (np.abs(ex.astype(np.float64) - true)[valid] <= 2.5e-5 * np.linalg.norm(x, axis=1).max() * np.linalg.norm(q, axis=1).max() + 1e-7).all(), name
        idx.close()


def test_query_overflowing_fp16_is_answered_exactly(rmu):
    """a query component >= ~1024 overflows fp16(64 q); such a query must be re-run on the exact scan."""
    x = O.make_corpus(20_000)
    q, _ = O.make_queries(x, 130)
    q = q.copy()
    q[5] *= 50_000.0                                              # |q_i| ~ 2500: fp16(64 q) = inf
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(q, 10)
    assert idx.last_screened() < 0
    os_, or_ = O.flat_search(q, x, 12)
    scale = np.maximum(1.0, np.abs(os_).max(axis=1, keepdims=True))
    for i in range(q.shape[0]):
        assert_topk_parity(s[i:i + 1], r[i:i + 1], os_[i:i + 1], or_[i:i + 1], score_tol=1e-4 * float(scale[i, 0]), tie_tol=1e-6 * float(scale[i, 0]))
    idx.close()


# ---- native squared-L2 metric (Milvus' default metric_type on raw vectors; RAGHelper.py:388-394) ------------------------------
@pytest.mark.parametrize("d,nq,k", [(384, 70, 10), (100, 33, 20), (384, 1, 10), (767, 9, 5), (200, 200, 40)])
def test_l2_metric_on_unnormalised_rows(rmu, d, nq, k):
    from ragmeup_amd import _native as N
    rng = np.random.default_rng(100 + d)
    x = (rng.standard_normal((12_345, d)) * rng.uniform(0.2, 4.0, (12_345, 1))).astype(np.float32)
    q = (x[rng.permutation(12_345)[:nq]] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = rmu.FlatIndex(d, metric=N.METRIC_L2SQ)
    idx.add(x[:5000]); idx.add(x[5000:])
    dist, r = idx.search(q, k)
    assert (np.diff(dist, axis=1) >= 0).all() and (dist >= 0).all()              # squared distances, nearest first
    os_, or_ = O.flat_search(q, x, k + 4, metric=O.METRIC_L2SQ)                   # oracle: -(|q - x|^2), larger = better
    scale = float(np.abs(os_).max())
    assert_topk_parity(-dist, r, os_, or_, score_tol=2e-6 * scale + 1e-4, tie_tol=1e-6 * scale)
    # IP ranking differs on these rows: the native metric is not the unit-norm shortcut
    ip = rmu.FlatIndex(d)
    ip.add(x)
    _, r_ip = ip.search(q, k)
    assert (r_ip != r).any()
    dead = np.unique(r[:, 0])
    idx.remove_rows(dead)
    dist2, r2 = idx.search(q, k)
    assert not np.isin(r2, dead).any()
    alive = np.ones(len(x), bool); alive[dead] = False
    assert_topk_parity(-dist2, r2, *O.flat_search(q, x, k + 4, metric=O.METRIC_L2SQ, alive=alive), score_tol=2e-6 * scale + 1e-4, tie_tol=1e-6 * scale)
    idx.close(); ip.close()


@pytest.mark.parametrize("n,nq,k", [(60_000, 200, 10), (60_000, 1024, 24), (60_000, 7, 32), (41_003, 130, 1), (300_000, 384, 10), (262_144 + 17, 129, 27)])
def test_l2_metric_on_the_screening_path(rmu, n, nq, k, tmp_path):
    """(round 5) The native L2 index at dim 384 screens like the inner-product index: the fp16 image scores q.x, the row's -|x|^2 / 2 starts
    the MFMA chain as its C operand (scan_screen_lean3_kernel<.., L2N = 1>), the fp32 re-score repeats the exact L2 scan's chain.  Raw vectors
    with norms 0.2 .. 12 (the ranking differs from the inner product's); distances and ids BIT-IDENTICAL to the exact L2 scan; ids equal
    to the fp64 oracle's under the tie rule; tombstones, growth in pieces, save / load (the image and the norms are rebuilt)."""
    from ragmeup_amd import _native as N
    rng = np.random.default_rng(500 + k)
    x = (rng.standard_normal((n, 384)) * rng.uniform(0.02, 0.6, (n, 1))).astype(np.float32)
    src = rng.permutation(n)[:nq]
    q = (x[src] + 0.002 * rng.standard_normal((nq, 384))).astype(np.float32)
    idx = rmu.FlatIndex(384, metric=N.METRIC_L2SQ, capacity_hint=64)
    idx.set_screen_min_batch(1)
    for lo in range(0, n, 25_000):
        idx.add(x[lo:lo + 25_000])
    dist, r = idx.search(q, k)
    assert idx.last_screened() != 0, "expected the screening path"
    if n >= 262_144:
        assert idx.last_geometry()["launches"] >= 3, "expected a multi-launch ladder"
    assert (np.diff(dist, axis=1) >= 0).all() and (dist >= 0).all()
    assert (r[:, 0] == src).all()
    idx.set_screening(False)
    d2, r2 = idx.search(q, k)
    assert idx.last_screened() == 0
    assert np.array_equal(r2, r) and np.array_equal(d2, dist)                      # bit-identical to the exact L2 scan
    idx.set_screening(True)
    sub = slice(0, 64)
    os_, or_ = O.flat_search(q[sub], x, k + 4, metric=O.METRIC_L2SQ)
    scale = float(np.abs(os_).max())
    assert_topk_parity(-dist[sub], r[sub], os_, or_, score_tol=2e-6 * scale + 1e-4, tie_tol=1e-6 * scale)
    ip = rmu.FlatIndex(384)
    ip.add(x)
    assert (ip.search(q, max(k, 10))[1][:, :k] != r).any()                          # not the inner product's ranking
    ip.close()
    dead = np.unique(r[:, 0])
    idx.remove_rows(dead)
    d3, r3 = idx.search(q, k)
    assert idx.last_screened() != 0 and not np.isin(r3, dead).any()
    idx.set_screening(False)
    d4, r4 = idx.search(q, k)
    assert np.array_equal(r4, r3) and np.array_equal(d4, d3)
    path = str(tmp_path / "l2s.rmu")
    idx.save(path)
    back = rmu.FlatIndex.load(path)
    back.set_screen_min_batch(1)
    d5, r5 = back.search(q, k)
    assert back.last_screened() != 0 and np.array_equal(r5, r3) and np.array_equal(d5, d3)
    idx.close(); back.close()


def test_l2_screening_falls_back_on_dense_ties(rmu):
    """41 copies of one row: the query next to it fails the sufficiency test and is answered by the exact L2 scan -- ascending row ids among
    the ties, distances as the exact path reports them; the other queries stay on the screened path."""
    from ragmeup_amd import _native as N
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((20_000, 384)) * rng.uniform(0.02, 0.3, (20_000, 1))).astype(np.float32)
    xd = np.concatenate([x, np.repeat(x[77:78], 40, axis=0)])
    q = np.concatenate([x[77:78], (x[rng.permutation(20_000)[:199]] + 0.01 * rng.standard_normal((199, 384))).astype(np.float32)])
    idx = rmu.FlatIndex(384, metric=N.METRIC_L2SQ)
    idx.add(xd)
    d, r = idx.search(q, 10)
    assert -8 <= idx.last_screened() <= -1            # the tied query for certain; short rows near the origin may flag a neighbour or two
    assert list(r[0]) == [77] + list(range(20_000, 20_009))
    os_, or_ = O.flat_search(q, xd, 14, metric=O.METRIC_L2SQ)
    # |q|^2 - (2 q.x - |x|^2) cancels at a duplicate: the fp32 chain's error scales with |q||x| (up to 35 here), not with the distance
    scale = float((q.astype(np.float64) ** 2).sum(1).max())
    assert_topk_parity(-d, r, os_, or_, score_tol=2e-4 * scale, tie_tol=2e-6 * scale)      # (384 roundings of 2^-24 |q||x| each way)
    idx.set_screening(False)
    d2, r2 = idx.search(q, 10)
    assert np.array_equal(r2, r) and np.array_equal(d2, d)
    idx.close()


def test_l2_metric_save_load_merge_and_limits(rmu, tmp_path):
    from ragmeup_amd import _native as N
    from ragmeup_amd.index import topk_merge
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((4000, 384)) * 2).astype(np.float32)
    q = rng.standard_normal((12, 384)).astype(np.float32)
    idx = rmu.FlatIndex(384, metric=N.METRIC_L2SQ)
    idx.add(x)
    d0, r0 = idx.search(q, 8)
    p = str(tmp_path / "l2.rmu")
    idx.save(p)
    back = rmu.FlatIndex.load(p)
    assert back.metric == N.METRIC_L2SQ                                         # read from the file header, not assumed
    d1, r1 = back.search(q, 8)
    assert np.array_equal(r0, r1) and np.array_equal(d0, d1)
    # two shards + merge with RMU_F_SMALLER_BETTER == the single index
    a, b = rmu.FlatIndex(384, metric=N.METRIC_L2SQ), rmu.FlatIndex(384, metric=N.METRIC_L2SQ)
    a.add(x[:1500]); b.add(x[1500:])
    da, ra = a.search(q, 8, row_base=0)
    db, rb = b.search(q, 8, row_base=1500)
    dm, rm = topk_merge(np.stack([da, db]), np.stack([ra, rb]), smaller_better=True)
    assert np.array_equal(rm, r0) and np.allclose(dm, d0, rtol=0, atol=0)
    with pytest.raises(N.RmuError):
        rmu.FlatIndex(768, metric=N.METRIC_L2SQ)                                # no spare column at 768
    for i in (idx, back, a, b):
        i.close()


def test_native_comm_world1(rmu, corpus50k):
    """The C-ABI exchange (rmu_comm_unique_id / rmu_comm_init / rmu_shard_allgather_topk: pack -> ONE ncclAllGather ->
    device merge inside librmu.so) at world size 1, the only size a 1-GPU box allows; 2 ranks: test_two_rank_gpu.py."""
    import torch
    from ragmeup_amd.shard import NativeComm, ShardedSearcher
    x, q, _ = corpus50k
    comm = NativeComm(NativeComm.unique_id(), 1, 0, device=0)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    ss = ShardedSearcher(idx, row_base=1000, comm=comm, force_collective=True)
    s, r = ss.search(torch.from_numpy(q[:96]).cuda(), 10)
    os_, or_ = O.flat_search(q[:96], x, 10)
    assert_topk_parity(s.cpu().numpy(), r.cpu().numpy() - 1000, os_, or_)
    # host-buffer form of the same entry point
    import ctypes
    from ragmeup_amd import _native as N
    ls, lr = idx.search(q[:7], 5)
    out_s, out_r = np.empty_like(ls), np.empty_like(lr)
    N.check(N.lib().rmu_shard_allgather_topk(comm._h, ls.ctypes.data, lr.ctypes.data, 7, 5, 0, out_s.ctypes.data, out_r.ctypes.data, 0),
            "rmu_shard_allgather_topk")
    assert np.array_equal(out_s, ls) and np.array_equal(out_r, lr)
    idx.close()
    # distance lists (RMU_METRIC_L2SQ): the searcher takes the merge direction from the index -- a larger-is-better merge
    # of even ONE ascending list would come back re-sorted the wrong way round
    rng = np.random.default_rng(3)
    xl = (rng.standard_normal((20_000, 384)) * rng.uniform(0.5, 2.0, (20_000, 1))).astype(np.float32)
    ql = xl[:40] + 0.05 * rng.standard_normal((40, 384)).astype(np.float32)
    il2 = rmu.FlatIndex(384, metric=N.METRIC_L2SQ)
    il2.add(xl)
    ss2 = ShardedSearcher(il2, row_base=0, comm=comm, force_collective=True)
    assert ss2.smaller_better
    d, r = ss2.search(torch.from_numpy(ql).cuda(), 10)
    d0, r0 = il2.search(ql, 10)
    assert np.array_equal(r.cpu().numpy(), r0) and np.allclose(d.cpu().numpy(), d0) and (np.diff(d0, axis=1) >= 0).all()
    comm.close(); il2.close()


def test_search_on_caller_stream_is_asynchronous_and_correct(rmu):
    """rmu.h stream contract: with a caller stream and device buffers the search (screening path included: the re-run
    decision is taken on the device) is only ORDERED on that stream; results are right after the caller synchronises."""
    import torch
    from ragmeup_amd import _native as N
    x = O.make_corpus(40_000)
    xd = np.concatenate([x, np.repeat(x[5:6], 40, axis=0)])                    # forces one re-run (41 copies of row 5)
    q = np.concatenate([x[5:6], O.make_queries(x, 255)[0]])
    idx = rmu.FlatIndex(384)
    idx.add(xd)
    st = torch.cuda.Stream()
    qd = torch.from_numpy(q).cuda()
    out_s = torch.empty((256, 10), dtype=torch.float32, device="cuda")
    out_r = torch.empty((256, 10), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    N.check(N.lib().rmu_index_search(idx._h, qd.data_ptr(), 256, 10, N.F_Q_DEVICE | N.F_OUT_DEVICE, 0, out_s.data_ptr(),
                                     out_r.data_ptr(), st.cuda_stream), "rmu_index_search")
    st.synchronize()
    assert_topk_parity(out_s.cpu().numpy(), out_r.cpu().numpy(), *O.flat_search(q, xd, 12))
    assert list(out_r[0].cpu().numpy()) == [5] + list(range(40_000, 40_009))
    idx.close()


def test_back_to_back_searches_on_different_caller_streams_do_not_share_scratch(rmu):
    """The per-thread scratch (query image, partial lists, thresholds, candidate keys) of an un-drained caller-stream
    search is still in use when the same thread issues the next search on ANOTHER stream (or stream 0): the library
    orders the second behind the first (rmu.h stream contract).  Without that the second call's memsets and partial
    lists land under the first call's kernels."""
    import torch
    from ragmeup_amd import _native as N
    x = O.make_corpus(300_000)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    qa, _ = O.make_queries(x, 512, seed=1)
    qb, _ = O.make_queries(x, 512, seed=2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    da, db = torch.from_numpy(qa).cuda(), torch.from_numpy(qb).cuda()
    outs = [(torch.empty((512, 10), dtype=torch.float32, device="cuda"), torch.empty((512, 10), dtype=torch.int64, device="cuda")) for _ in range(3)]
    torch.cuda.synchronize()
    lib = N.lib()
    fl = N.F_Q_DEVICE | N.F_OUT_DEVICE
    for rep in range(3):
        N.check(lib.rmu_index_search(idx._h, da.data_ptr(), 512, 10, fl, 0, outs[0][0].data_ptr(), outs[0][1].data_ptr(), sa.cuda_stream), "a")
        N.check(lib.rmu_index_search(idx._h, db.data_ptr(), 512, 10, fl, 0, outs[1][0].data_ptr(), outs[1][1].data_ptr(), sb.cuda_stream), "b")
        s0, r0 = idx.search(qa[:64], 10)                                      # stream 0 (internal), host buffers, drained
        sa.synchronize(); sb.synchronize()
        assert_topk_parity(outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy(), *O.flat_search(qa, x, 12))
        assert_topk_parity(outs[1][0].cpu().numpy(), outs[1][1].cpu().numpy(), *O.flat_search(qb, x, 12))
        assert_topk_parity(s0, r0, *O.flat_search(qa[:64], x, 12))
    idx.close()


# ---- deep k over a large corpus (BASELINE config 5: dense top-100): the exact 128-deep scan run as a threshold ladder over
# growing row ranges (>= 262144 rows; below that, and with RMU_DEEP=0, the single cold launch) -----------------------------------
@pytest.fixture(scope="module")
def corpus300k():
    x = O.make_corpus(300_000, seed=77)
    q, planted = O.make_queries(x, 130, seed=78)
    return x, q, planted


@pytest.mark.parametrize("nq,k", [(64, 100), (5, 33), (33, 112), (130, 100), (1, 64)])
def test_deep_k_ladder_parity_vs_oracle(rmu, corpus300k, nq, k):
    """The EXACT 128-deep ladder (k > 104 by default; k > 32 with the screening switched off, which is how it is reached here)."""
    x, q, planted = corpus300k
    idx = rmu.FlatIndex(384)
    idx.add(x)
    idx.set_screening(False)
    s, r = idx.search(q[:nq], k)
    assert idx.last_screened() == 0
    assert idx.last_geometry()["launches"] >= 2                    # really the ladder (one scan launch per row range)
    assert_topk_parity(s, r, *O.flat_search(q[:nq], x, k + 4))
    assert (r[:, 0] == planted[:nq]).all() and (np.diff(s, axis=1) <= 0).all()
    # device buffers give the same bits
    import torch
    sd, rd = idx.search(torch.from_numpy(q[:nq]).cuda(), k)
    assert np.array_equal(rd.cpu().numpy(), r) and np.array_equal(sd.cpu().numpy(), s)
    # ... and so does the single cold launch over a prefix that is too small for the ladder (same rows, same arithmetic)
    small = rmu.FlatIndex(384)
    small.add(x[:200_000])
    small.set_screening(False)
    s1, r1 = small.search(q[:nq], k)
    assert small.last_geometry()["launches"] == 1
    assert_topk_parity(s1, r1, *O.flat_search(q[:nq], x[:200_000], k + 4))
    small.close(); idx.close()


@pytest.mark.parametrize("nq,k", [(64, 100), (5, 33), (130, 64), (1024, 100), (1, 104), (200, 40), (33, 112)])
def test_deep_k_screening_is_bit_identical_to_the_exact_ladder(rmu, corpus300k, nq, k):
    """(round 6; BASELINE config 5's dense top-100, server/RAGHelper.py:497-499 with fetch_k = 100) 32 < k <= 104 over >= 262144 rows
    takes the fp16 screening ladder with K' = k + max(8, k / 5) <= 120 candidates (slots of 128 keys, two keys per lane in the
    compaction / emit / re-score) + the exact fp32 re-score: ids AND scores bit-identical to the exact fp32 128-deep ladder, for one-tile
    (nq <= 128) and full-tile geometries; k > 104 stays on the exact ladder."""
    x, q, planted = corpus300k
    qq = q[:nq] if nq <= q.shape[0] else np.concatenate([q] * ((nq + q.shape[0] - 1) // q.shape[0]))[:nq]
    idx = rmu.FlatIndex(384)
    idx.add(x)
    s, r = idx.search(qq, k)
    if k <= 104:
        assert idx.last_screened() != 0, "expected the screening path"
    else:
        assert idx.last_screened() == 0
    idx.set_screening(False)
    s2, r2 = idx.search(qq, k)
    assert idx.last_screened() == 0
    assert np.array_equal(r2, r) and np.array_equal(s2, s)
    sub = slice(0, min(nq, 16))
    assert_topk_parity(s[sub], r[sub], *O.flat_search(qq[sub], x, k + 4))
    idx.close()


def test_deep_k_screening_on_the_l2_index_and_with_near_duplicates(rmu, corpus300k):
    """k = 100 on the NATIVE squared-L2 index (Milvus' default metric, server/RAGHelper.py:388-394) screens like the inner-product
    index; 100 near-duplicates in consecutive rows (every candidate in ONE chunk's slot: the compaction path with two keys per lane) and
    deleted rows."""
    from ragmeup_amd import _native as N
    x, q, _ = corpus300k
    rng = np.random.default_rng(6)
    dup = x[1000][None, :] + 1e-3 * rng.standard_normal((150, 384)).astype(np.float32)
    dup /= np.linalg.norm(dup, axis=1, keepdims=True)
    xd = np.concatenate([x[:150_000], dup, x[150_000:]])
    qq = np.concatenate([x[1000][None, :], dup[:40], q[:24]])
    for metric in (N.METRIC_IP, N.METRIC_L2SQ):
        idx = rmu.FlatIndex(384, metric)
        idx.add(xd)
        s, r = idx.search(qq, 100)
        assert idx.last_screened() != 0
        idx.set_screening(False)
        s2, r2 = idx.search(qq, 100)
        assert idx.last_screened() == 0 and np.array_equal(r2, r) and np.array_equal(s2, s)
        idx.set_screening(True)
        assert set(r[0].tolist()) <= set(range(150_000, 150_150)) | {1000}
        dead = np.arange(0, xd.shape[0], 3)
        idx.remove_rows(dead)
        s, r = idx.search(qq[:20], 100)
        assert not np.isin(r, dead).any()
        idx.set_screening(False)
        s2, r2 = idx.search(qq[:20], 100)
        assert np.array_equal(r2, r) and np.array_equal(s2, s)
        idx.close()


def test_deep_k_ladder_with_clustered_duplicates_tombstones_and_l2(rmu, corpus300k):
    """100 near-duplicates of one row stored in consecutive rows (one corpus part, one ladder range) must all come back; deleted
    rows never do; the native-L2 index takes the same ladder."""
    from ragmeup_amd import _native as N
    x, q, _ = corpus300k
    rng = np.random.default_rng(5)
    dup = x[1000][None, :] + 1e-3 * rng.standard_normal((100, 384)).astype(np.float32)
    dup /= np.linalg.norm(dup, axis=1, keepdims=True)
    xd = np.concatenate([x[:150_000], dup, x[150_000:]])
    qq = np.concatenate([x[1000][None, :], dup[:40], q[:8]])
    idx = rmu.FlatIndex(384)
    idx.add(xd)
    s, r = idx.search(qq, 100)
    assert_topk_parity(s, r, *O.flat_search(qq, xd, 104))
    assert set(r[0].tolist()) <= set(range(150_000, 150_100)) | {1000}
    dead = np.arange(0, xd.shape[0], 3)
    idx.remove_rows(dead)
    alive = np.ones(xd.shape[0], bool); alive[dead] = False
    s, r = idx.search(qq[:20], 100)
    assert_topk_parity(s, r, *O.flat_search(qq[:20], xd, 104, alive=alive))
    idx.close()
    xl = (rng.standard_normal((270_000, 384)) * rng.uniform(0.5, 2.0, (270_000, 1))).astype(np.float32)
    ql = xl[:12] + 0.05 * rng.standard_normal((12, 384)).astype(np.float32)
    il2 = rmu.FlatIndex(384, metric=N.METRIC_L2SQ)
    il2.add(xl)
    d, r = il2.search(ql, 64)
    os_, or_ = O.flat_search(ql, xl, 68, metric=O.METRIC_L2SQ)
    scale = float((xl ** 2).sum(1).max())
    assert_topk_parity(-d, r, os_, or_, score_tol=2e-6 * scale + 1e-4, tie_tol=1e-6 * scale)
    il2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"RMU_SCREEN_PACE": "0"}, {"RMU_NT": "0"}, {"RMU_NO_SHARED_THR": "1"}],
                         ids=["default_lean3", "no_sibling_pacing", "no_nontemporal_stream", "no_shared_thresholds"])
def test_every_switchable_screening_kernel_returns_the_exact_answers(env):
    """Every form of the screening scan the PRODUCT library can be switched to (environment, read once per process) answers the same
    batches -- full query tiles, a ragged tile, one query tile, a lone wave -- and every answer must be the exact fp32 scan's, bit for
    bit, through the screening path.  Round 5: the product library carries scan_screen_lean3_kernel only (tests/test_abi_cpu.py checks the
    symbol table); round 3's kernel, the lean / lean2 steps, the K-split and 128-queries-per-wave forms exist in debug builds
    (python -m ragmeup_amd.build --debug-kernels), where RMU_SCREEN_LEAN / _LEAN4 / _W8 / _KS / _G4 select them: profiles/r04_ab_screen_forms.txt."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "screen_variant_driver.py")], env=dict(os.environ, RMU_TUNING="1", **env),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for nq, r in res.items():
        assert r["screened"] == 1 and r["same"], (env, nq, r)


@pytest.mark.gpu
def test_repeated_searches_into_caller_owned_tensors_stay_exact(rmu):
    """FlatIndex.search(out=(scores, rows)): a serving loop's pre-allocated result tensors.  Repeated searches with identical addresses must
    return what the first returned, see rows deleted in between (tombstones are poisoned in place), other query VALUES in the same tensor,
    and a grown index -- always the exact scan's bits.  (Round 5 also captured such searches into a hipGraph and replayed them: bit-identical,
    and measured NO faster -- 1.387 vs 1.372 ms at batch 32 over 10M rows, profiles/r05_search_graph_ab.txt -- so the capture was removed.)"""
    import torch
    x = O.make_corpus(300_000, seed=31)
    q, planted = O.make_queries(x, 200, seed=32)
    idx = rmu.FlatIndex(384)
    idx.add(x)
    qd = torch.from_numpy(q).cuda()
    out = (torch.empty((200, 10), dtype=torch.float32, device="cuda"), torch.empty((200, 10), dtype=torch.int64, device="cuda"))
    first = None
    for rep in range(5):                                        # eager, capture + launch, replay x 3
        s, r = idx.search(qd, 10, out=out)
        assert idx.last_screened() != 0
        got = (s.cpu().numpy().copy(), r.cpu().numpy().copy())
        if first is None:
            first = got
            assert (got[1][:, 0] == planted).all()
            assert_topk_parity(got[0][:64], got[1][:64], *O.flat_search(q[:64], x, 14))
        assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), rep
    # other query VALUES in the same tensor: the replay reads the tensor, not a copy
    q2, planted2 = O.make_queries(x, 200, seed=33)
    qd.copy_(torch.from_numpy(q2))
    s, r = idx.search(qd, 10, out=out)
    assert (r[:, 0].cpu().numpy() == planted2).all()
    s_ex, r_ex = idx.search(qd, 112)                            # exact scan (k > 104)
    assert torch.equal(r_ex[:, :10], r) and torch.equal(s_ex[:, :10], s)
    # rows deleted between two replays of the same graph
    dead = np.unique(planted2[:50])
    idx.remove_rows(dead)
    s, r = idx.search(qd, 10, out=out)
    assert not np.isin(r.cpu().numpy(), dead).any()
    alive = np.ones(len(x), bool); alive[dead] = False
    assert_topk_parity(s.cpu().numpy()[:48], r.cpu().numpy()[:48], *O.flat_search(q2[:48], x, 14, alive=alive))
    # the index grows: another key, the new rows are found
    extra = q2[:20] / np.linalg.norm(q2[:20], axis=1, keepdims=True)
    first_new = idx.add(extra.astype(np.float32))
    for rep in range(3):
        s, r = idx.search(qd, 10, out=out)
        assert (r[:20, 0].cpu().numpy() == first_new + np.arange(20)).all(), rep
    # deep k (the exact threshold ladder) through the same mechanism
    outd = (torch.empty((64, 100), dtype=torch.float32, device="cuda"), torch.empty((64, 100), dtype=torch.int64, device="cuda"))
    ref = None
    for rep in range(4):
        s, r = idx.search(qd[:64], 100, out=outd)
        cur = (s.cpu().numpy().copy(), r.cpu().numpy().copy())
        ref = ref or cur
        assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])
    with pytest.raises(ValueError):
        idx.search(qd, 10, out=(out[0][:10], out[1]))
    idx.close()



def test_a_torch_graph_capture_on_another_thread_survives_growth_removal_search_and_free(rmu):
    """VERDICT r5 next-4 / SURVEY 8b ("all entry points thread-safe"), server/RAGHelper_local.py:42-105: the reference's LLM runs in PyTorch in
    the same process and on the same GPU; a torch.cuda.graph capture there (GLOBAL capture mode) is invalidated by a hipDeviceSynchronize or a
    synchronous hipMemcpy on ANY thread, and by most synchronous HIP calls of a thread in the default capture-interaction mode
    (profiles/r06_capture_probe.txt).  The library now waits for exactly the streams that used a buffer and runs every entry point in relaxed
    mode: while thread A is INSIDE a capture, thread B creates an index, appends (several re-allocations), searches on its internal stream and
    on a caller stream left in flight, tombstones rows, saves / loads and frees -- ten rounds; every capture instantiates and replays with
    the right numbers and every search is exact."""
    import ctypes
    import tempfile
    import torch
    from ragmeup_amd import _native as N
    x = O.make_corpus(24_000)
    q, _ = O.make_queries(x, 32)
    want = O.flat_search(q, x, 12)
    # everything thread B needs from torch exists BEFORE the captures start (a torch allocation or H2D copy on B would be torch's own
    # interference with torch's capture, not this library's)
    qd = torch.from_numpy(q).cuda()
    out_s = torch.empty((32, 10), dtype=torch.float32, device="cuda")
    out_r = torch.empty((32, 10), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    a = torch.randn((256, 256), device="cuda")
    b = torch.randn((256, 256), device="cuda")
    ref = (a * b + 1.0).relu().cpu()        # (elementwise only: a first rocBLAS call inside a capture on a fresh thread crashes in torch itself)
    torch.cuda.synchronize()
    errs = []
    tmp = tempfile.mkdtemp()
    for rnd in range(10):
        inside, proceed = threading.Event(), threading.Event()

        def capturer():
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):                  # capture_error_mode="global" (the default)
                    y = a * b
                    inside.set()
                    assert proceed.wait(120)
                    y = (y + 1.0).relu()
                g.replay()
                torch.cuda.synchronize()
                if not torch.allclose(y.cpu(), ref, atol=1e-3):
                    errs.append(("capture replay wrong", rnd))
            except Exception as e:   # noqa: BLE001
                errs.append(("capturer", rnd, repr(e)))
            finally:
                inside.set()

        def library_user():
            try:
                assert inside.wait(120)
                idx = rmu.FlatIndex(384, capacity_hint=16)                       # hipMalloc + zero fill
                for lo in range(0, 24_000, 3000):                                # re-allocations: old matrices freed behind reader events
                    idx.add(x[lo:lo + 3000])
                    if lo == 9000:                                               # a search left IN FLIGHT on a caller stream, then growth behind it
                        N.check(N.lib().rmu_index_search(idx._h, qd.data_ptr(), 32, 10, N.F_Q_DEVICE | N.F_OUT_DEVICE, 0, out_s.data_ptr(),
                                                         out_r.data_ptr(), side.cuda_stream), "rmu_index_search")
                assert idx.stats()["grow_count"] >= 5
                s, r = idx.search(q, 10)                                         # internal stream, host buffers
                assert_topk_parity(s, r, *want)
                dead = np.unique(r[:, 0])
                assert idx.remove_rows(dead) == dead.size
                s2, r2 = idx.search(q, 10)
                assert not np.isin(r2, dead).any()
                p = os.path.join(tmp, f"cap{rnd}.rmu")
                idx.save(p)
                idx.close()                                                      # hipFree x3
                back = rmu.FlatIndex.load(p)
                s3, r3 = back.search(q, 10)
                assert np.array_equal(r3, r2) and np.array_equal(s3, s2)
                back.close()
            except Exception as e:   # noqa: BLE001
                errs.append(("library", rnd, repr(e)))
            finally:
                proceed.set()

        ts = [threading.Thread(target=capturer), threading.Thread(target=library_user)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
    side.synchronize()
    # the search that was left in flight over 12 000 rows while the matrix was re-allocated under it read the OLD matrix to the end
    assert_topk_parity(out_s.cpu().numpy(), out_r.cpu().numpy(), *O.flat_search(q, x[:12_000], 12))
