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
def test_flush_on_the_queues_own_stream_matches_oracle():
    """matinv_queue_stream: submit + flush inside the queue's own stream (only the chain of the largest bin forks), two queues
    alternating as bench.py's mixed workload runs them; the caller joins with wait_stream."""
    import torch
    bq = pkg("binqueue")
    rng = np.random.default_rng(23)
    sizes = [32, 300, 128, 20, 512, 64, 33, 200]
    qs = [bq.SizeBinnedQueue(), bq.SizeBinnedQueue()]
    homes = [q.home_stream(torch.float64) for q in qs]
    assert homes[0].cuda_stream != homes[1].cuda_stream and homes[0].cuda_stream != 0
    results, wants = [], []
    for rnd in range(4):
        q, home = qs[rnd % 2], homes[rnd % 2]
        want, items = [], []
        for i, n in enumerate(sizes):
            B = spd_batch(n, 1, seed=1000 * rnd + i)
            a, c, d = (rng.random(n) for _ in range(3))
            want.append(oracle.mean_batched(a, B, c, d, n)[0])
            items.append([torch.from_numpy(x).cuda() for x in (a, B, c, d)])
        torch.cuda.synchronize()  # the items were written on the default stream
        with torch.cuda.stream(home):
            for it in items:
                q.submit(*it)
            m, v = q.flush()
        assert v is None
        results.append(m)
        wants.append(want)
    for home in homes:
        torch.cuda.current_stream().wait_stream(home)
    for m, want in zip(results, wants):
        assert np.abs(m.cpu().numpy() - np.array(want)).max() < 1e-10


@pytest.mark.gpu
def test_device_side_padding_agrees_with_explicit_padding():
    """The queue never pads in memory (the kernels pad to their tile size in registers); an explicitly identity-padded copy
    of an item -- the reference sketch's pad-to-the-bin policy -- gives the same scalar."""
    import torch
    bq = pkg("binqueue")
    rng = np.random.default_rng(9)
    sizes = [5, 17, 40, 100, 128, 33, 130, 300, 16, 64]
    qa, qb = bq.SizeBinnedQueue(), bq.SizeBinnedQueue()
    want = []
    for i, n in enumerate(sizes):
        B = spd_batch(n, 1, seed=200 + i)
        a, c, d = (rng.random(n) for _ in range(3))
        want.append(oracle.mean_batched(a, B, c, d, n)[0])
        t = [torch.from_numpy(x).cuda() for x in (a, B, c, d)]
        qa.submit(*t)
        qb.submit(*bq.pad_item(*t, n, bq.bin_of(n)))
    assert qa.pending() == qb.pending()
    ma, mb = qa.flush()[0].cpu().numpy(), qb.flush()[0].cpu().numpy()
    assert np.abs(ma - np.array(want)).max() < 1e-10 and np.abs(mb - np.array(want)).max() < 1e-10
    with pytest.raises(ValueError):
        qa.submit(*[torch.zeros(k, dtype=torch.float64, device="cuda") for k in (2000, 2000 * 2000, 2000, 2000)])


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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_chunks_runs_and_shuffled_tickets(dtype):
    """Chunks of one big allocation submitted in shuffled order (one contiguous run, tickets out of order), chunks from
    separate allocations (gather), single chunks (zero copy), several n per bin; with and without variances."""
    import torch
    bq = pkg("binqueue")
    dt = getattr(torch, dtype)
    rng = np.random.default_rng(5)
    tol = 1e-10 if dtype == "float64" else 5e-5
    for with_e in (True, False):
        q = bq.SizeBinnedQueue()
        want = {}
        keep = []
        chunk_log, ticket_log = [], []

        def add_group(n, count, pieces, shuffle):
            B = spd_batch(n, count, seed=n + count)
            a, c, d = (rng.random(count * n) for _ in range(3))
            e = rng.random(count)
            m = oracle.mean_batched(a, B, c, d, n)
            v = oracle.variance_batched(a, B, c, e, n)
            ta, tB, tc, td, te = (torch.from_numpy(x).to(dt).cuda() for x in (a, B, c, d, e))
            keep.append((ta, tB, tc, td, te))
            cuts = np.linspace(0, count, pieces + 1).astype(int)
            order = list(range(pieces))
            if shuffle:
                order = order[::-1]
            for k in order:
                lo, hi = int(cuts[k]), int(cuts[k + 1])
                if hi == lo:
                    continue
                ch = (ta[lo * n:hi * n], tB[lo * n * n:hi * n * n], tc[lo * n:hi * n], td[lo * n:hi * n]) + ((te[lo:hi],) if with_e else ())
                t = q.submit_many(n, *ch)
                chunk_log.append(ch)
                ticket_log.append(t)
                for i in range(hi - lo):
                    want[t + i] = (m[lo + i], v[lo + i])

        add_group(32, 300, 5, True)     # one run, tickets reversed
        add_group(20, 64, 1, False)     # same bin, other n: its own launch, zero copy
        add_group(128, 40, 4, False)    # one run, tickets in order
        add_group(100, 17, 3, True)
        add_group(32, 100, 2, False)    # second allocation with n = 32: two runs -> gathered
        add_group(300, 3, 1, False)
        add_group(7, 9, 3, True)
        means, variances = q.flush()
        torch.cuda.synchronize()
        assert (variances is not None) == with_e
        # the same chunks again through ONE C call (matinv_queue_submit_chunks)
        table = q.chunk_table([(int(ts[1].numel() // ts[0].numel()), *ts) for ts in chunk_log])
        firsts = q.submit_table(table)
        assert firsts == ticket_log
        m2, v2 = q.flush()
        torch.cuda.synchronize()
        assert torch.equal(m2, means) and (not with_e or torch.equal(v2, variances))
        total = len(want)
        assert means.numel() == total
        wm = np.array([want[i][0] for i in range(total)])
        assert np.abs(means.double().cpu().numpy() - wm).max() < tol
        if with_e:
            wv = np.array([want[i][1] for i in range(total)])
            assert np.abs(variances.double().cpu().numpy() - wv).max() < tol
        assert q.pending() == {}
