"""Size-binned multi-queue (README.md:41-44 of the reference; BASELINE configs[4])."""
import numpy as np
import pytest

import oracle
from conftest import pkg, spd_batch


def test_bin_and_shard_logic_cpu():
    bq = pkg("binqueue")
    assert [bq.bin_of(n) for n in (1, 32, 33, 128, 129, 512, 513, 1024)] == [32, 32, 128, 128, 512, 512, 1024, 1024]
    with pytest.raises(ValueError):
        bq.bin_of(1025)
    sizes = [32] * 10 + [1024] * 3 + [128] * 7 + [512] * 4
    parts = [bq.shard_items(sizes, r, 4) for r in range(4)]
    assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
    cost = [sum(sizes[i] ** 3 for i in p) for p in parts]
    assert max(cost) - min(cost) <= 1024 ** 3  # within one largest item


def test_padding_leaves_the_scalar_unchanged_cpu():
    import torch
    bq = pkg("binqueue")
    n, nb = 5, 8
    rng = np.random.default_rng(0)
    B = spd_batch(n, 1, seed=3)
    a, c, d = (rng.random(n) for _ in range(3))
    t = [torch.from_numpy(x) for x in (a, B, c, d)]
    ap, Bp, cp, dp = (x.numpy() for x in bq.pad_item(*t, n, nb))
    m0 = oracle.mean_batched(a, B, c, d, n)[0]
    m1 = oracle.mean_batched(ap, Bp, cp, dp, nb)[0]
    assert abs(m0 - m1) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_mixed_size_stream_matches_oracle(dtype):
    import torch
    bq = pkg("binqueue")
    dt = getattr(torch, dtype)
    rng = np.random.default_rng(11)
    sizes = [32, 7, 128, 100, 32, 512, 33, 64, 128, 16, 300, 32, 1024 if dtype == "float32" else 200]
    q = bq.SizeBinnedQueue()
    want_m, want_v = [], []
    for i, n in enumerate(sizes):
        B = spd_batch(n, 1, seed=100 + i)
        a, c, d = (rng.random(n) for _ in range(3))
        e = rng.random(1)
        want_m.append(oracle.mean_batched(a, B, c, d, n)[0])
        want_v.append(oracle.variance_batched(a, B, c, e, n)[0])
        to = lambda x: torch.from_numpy(x).to(dt).cuda()
        assert q.submit(to(a), to(B), to(c), to(d), to(e)) == i
    assert q.pending() == {32: 5, 128: 5, 512: 2, 1024: 1} if dtype == "float32" else True
    m, v = q.flush()
    torch.cuda.synchronize()
    tol = 1e-10 if dtype == "float64" else 5e-5
    assert np.abs(m.double().cpu().numpy() - np.array(want_m)).max() < tol
    assert np.abs(v.double().cpu().numpy() - np.array(want_v)).max() < tol
    assert q.pending() == {} and q.flush()[0].numel() == 0


@pytest.mark.gpu
def test_padding_policies_agree():
    """pad_to="tile" (groups by n rounded up to 16 inside a bin) and pad_to="bin" (the literal policy) give the same scalars."""
    import torch
    bq = pkg("binqueue")
    rng = np.random.default_rng(9)
    sizes = [5, 17, 40, 100, 128, 33, 130, 300, 16, 64]
    qa, qb = bq.SizeBinnedQueue(pad_to="tile"), bq.SizeBinnedQueue(pad_to="bin")
    want = []
    for i, n in enumerate(sizes):
        B = spd_batch(n, 1, seed=200 + i)
        a, c, d = (rng.random(n) for _ in range(3))
        want.append(oracle.mean_batched(a, B, c, d, n)[0])
        t = [torch.from_numpy(x).cuda() for x in (a, B, c, d)]
        qa.submit(*t)
        qb.submit(*t)
    ma, mb = qa.flush()[0].cpu().numpy(), qb.flush()[0].cpu().numpy()
    assert np.abs(ma - np.array(want)).max() < 1e-10 and np.abs(mb - np.array(want)).max() < 1e-10
    with pytest.raises(ValueError):
        bq.SizeBinnedQueue(pad_to="none")


@pytest.mark.gpu
def test_submit_many_matches_single_submits():
    """Chunked submission (same-size items back to back) gives the same results, in ticket order, as item-by-item."""
    import torch
    bq = pkg("binqueue")
    rng = np.random.default_rng(5)
    groups = [(20, 7), (32, 5), (100, 3), (20, 2), (130, 2)]  # (n, count): padded and exact bins, a bin hit twice
    q1, q2 = bq.SizeBinnedQueue(), bq.SizeBinnedQueue()
    want = []
    for gi, (n, cnt) in enumerate(groups):
        B = spd_batch(n, cnt, seed=50 + gi)
        a, c, d = (rng.random(cnt * n) for _ in range(3))
        e = rng.random(cnt)
        want += list(oracle.variance_batched(a, B, c, e, n))
        to = lambda x: torch.from_numpy(x).cuda()
        ta, tB, tc, td, te = (to(x) for x in (a, B, c, d, e))
        first = q1.submit_many(n, ta, tB, tc, td, te)
        assert first == len(want) - cnt
        for i in range(cnt):
            q2.submit(ta[i * n:(i + 1) * n], tB[i * n * n:(i + 1) * n * n], tc[i * n:(i + 1) * n], td[i * n:(i + 1) * n], te[i:i + 1])
    assert q1.pending() == q2.pending() == {32: 14, 128: 3, 512: 2}
    (m1, v1), (m2, v2) = q1.flush(), q2.flush()
    torch.cuda.synchronize()
    assert torch.equal(m1, m2) and torch.equal(v1, v2)
    assert np.abs(v1.cpu().numpy() - np.array(want)).max() < 1e-10
    with pytest.raises(ValueError):
        q1.submit_many(4, ta[:8], tB[:32], tc[:8], td[:7])
