"""Size-binned multi-queue for mixed-size Gaussian-process items.

The reference only sketches this ("use multiple queues for different sizes: 32, 128, 512, 1024",
/root/reference/README.md:41-44) and never built it; BASELINE.json configs[4] asks for it. Items (a, B, c, d[, e]) of
arbitrary n are appended to the queue of the smallest bin that holds them; `flush()` turns every non-empty queue into
device-resident batches -- one per padded size, see SizeBinnedQueue; padding is an identity block in B and zeros in the
vectors, which leaves a^T (B + diag c)^-1 d unchanged -- and launches the fused pipeline kernel on the bin's own HIP
stream, so the bins overlap on the device. Results come back in submission order.

torch supplies device memory and streams; all arithmetic is libmatinv_hip.so (api.calcluateMean / calcluateVariance).
Across GPUs the item list is dealt out by `shard_items` (largest first, round-robin) with no communication; rank results
are small (one scalar per item) and can be gathered with torch.distributed.all_gather_object by the caller.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

DEFAULT_BINS = (32, 128, 512, 1024)


def bin_of(n: int, bins: Sequence[int] = DEFAULT_BINS) -> int:
    """Smallest bin size >= n."""
    for b in bins:
        if n <= b:
            return b
    raise ValueError(f"n={n} exceeds the largest bin {bins[-1]}")


def shard_items(sizes: Sequence[int], rank: int, world: int) -> List[int]:
    """Indices of the items rank `rank` of `world` processes: items sorted by size (largest first, cost ~ n^3) are dealt
    round-robin, which balances the cubic cost to within one item per bin."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    return sorted(order[rank::world])


def pad_item(a, B, c, d, n: int, nb: int):
    """Embed an n-item into an nb-item: B -> blockdiag(B, I), vectors zero-extended. B is n*n column-major flat or (n, n)."""
    return pad_items(a, B, c, d, n, nb, 1)


def pad_items(a, B, c, d, n: int, nb: int, count: int):
    """`count` equally sized n-items (flat, item-major) embedded into nb-items in one shot."""
    import torch
    if n == nb:
        return a.reshape(-1), B.reshape(-1), c.reshape(-1), d.reshape(-1)
    Bp = torch.eye(nb, dtype=B.dtype, device=B.device).repeat(count, 1, 1)
    Bp[:, :n, :n] = B.reshape(count, n, n)

    def z(v):
        out = torch.zeros((count, nb), dtype=v.dtype, device=v.device)
        out[:, :n] = v.reshape(count, n)
        return out.reshape(-1)
    return z(a), Bp.reshape(-1), z(c), z(d)


class SizeBinnedQueue:
    """`pad_to="tile"` (default): inside a bin, items are grouped by n rounded up to a multiple of 16 (never beyond the bin) and
    each group is one launch at that size -- the kernels serve any n, so an n = 40 item in the 128 bin costs what a 48 x 48
    item costs, not what a 128 x 128 one does. `pad_to="bin"`: the reference sketch's literal policy, everything padded to
    the bin size, one launch per bin."""

    def __init__(self, bins: Sequence[int] = DEFAULT_BINS, device=None, pad_to: str = "tile"):
        import torch
        if pad_to not in ("tile", "bin"):
            raise ValueError("pad_to must be 'tile' or 'bin'")
        self.pad_to = pad_to
        self.bins = tuple(sorted(bins))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._queues = {b: [] for b in self.bins}   # bin -> list of chunks (first ticket, count, n, a, B, c, d, e)
        self._streams = {b: torch.cuda.Stream(device=self.device) for b in self.bins}
        self._tickets = 0

    def submit(self, a, B, c, d, e=None) -> int:
        """Queue one item; tensors are CUDA tensors of one dtype. Returns its ticket (position in the result)."""
        return self.submit_many(a.numel(), a, B, c, d, e, count=1)

    def submit_many(self, n: int, As, Bs, Cs, Ds, Es=None, count: Optional[int] = None) -> int:
        """Queue `count` items of one size n that already lie back to back (As: count*n, Bs: count*n*n column-major, ...;
        Es: count scalars or None). Returns the ticket of the first; the rest follow consecutively. No per-item host work."""
        if count is None:
            count = As.numel() // n
        if As.numel() != count * n or Bs.numel() != count * n * n or Cs.numel() != count * n or Ds.numel() != count * n \
                or (Es is not None and Es.numel() != count):
            raise ValueError("inconsistent item shapes")
        t = self._tickets
        self._tickets += count
        self._queues[bin_of(n, self.bins)].append((t, count, n, As, Bs, Cs, Ds, Es))
        return t

    def pending(self):
        return {b: sum(ch[1] for ch in q) for b, q in self._queues.items() if q}

    def flush(self) -> Tuple["torch.Tensor", Optional["torch.Tensor"]]:
        """Run every bin; returns (means, variances or None) in ticket order and empties the queues."""
        import torch
        from . import api
        total = self._tickets
        if total == 0:
            return torch.empty(0, device=self.device), None
        first = next(q[0] for q in self._queues.values() if q)
        dtype = first[3].dtype
        means = torch.empty(total, dtype=dtype, device=self.device)
        want_var = any(ch[7] is not None for q in self._queues.values() for ch in q)
        variances = torch.empty(total, dtype=dtype, device=self.device) if want_var else None
        cur = torch.cuda.current_stream(self.device)
        for b in sorted(self._queues, reverse=True):  # largest bin first: its many small launches overlap the rest
            q = self._queues[b]
            if not q:
                continue
            s = self._streams[b]
            s.wait_stream(cur)  # the items were produced on the caller's stream
            with torch.cuda.stream(s):
                groups = {}
                for ch in q:
                    pn = b if self.pad_to == "bin" else min(b, -(-ch[2] // 16) * 16)
                    groups.setdefault(pn, []).append(ch)
                for pn in sorted(groups, reverse=True):
                    g = groups[pn]
                    padded = [pad_items(a, B, c, d, n, pn, cnt) for (_, cnt, n, a, B, c, d, _) in g]
                    A_, B_, C_, D_ = (torch.cat([p[k] for p in padded]).contiguous() for k in range(4))
                    nitems = sum(ch[1] for ch in g)
                    idx = torch.cat([torch.arange(t, t + cnt, device=self.device) for (t, cnt, *_rest) in g]) \
                        if len(g) < 64 else torch.tensor([t + i for (t, cnt, *_rest) in g for i in range(cnt)], device=self.device)
                    out = api.calcluateMean(pn, A_, B_, C_, D_, batchSize=nitems)
                    means.index_copy_(0, idx, out)
                    if want_var:
                        E_ = torch.cat([(ch[7].reshape(-1) if ch[7] is not None
                                         else torch.zeros(ch[1], dtype=dtype, device=self.device)) for ch in g])
                        var = api.calcluateVariance(pn, A_, B_, C_, E_, batchSize=nitems)
                        variances.index_copy_(0, idx, var)
        for s in self._streams.values():
            cur.wait_stream(s)
        self._queues = {b: [] for b in self.bins}
        self._tickets = 0
        return means, variances
