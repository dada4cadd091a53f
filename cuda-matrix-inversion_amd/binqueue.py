"""Size-binned multi-queue for mixed-size Gaussian-process items: thin mirror of the C queue of libmatinv_hip.so.

The reference only sketches this ("use multiple queues for different sizes: 32, 128, 512, 1024",
/root/reference/README.md:41-44) and never built it; BASELINE.json configs[4] asks for it. The queue itself --
binning, grouping by exact n, run detection, the segmented gather, one stream per bin -- is C behind the C ABI
(csrc/queue.hip: matinv_queue_create / submit / flush, include/matinv.h); this module only keeps the submitted torch
tensors alive until the flush and hands their device pointers over. Results come back in submission (ticket) order.

Across GPUs the item list is dealt out by `shard_items` (largest first, round-robin) with no communication; rank results
are small (one scalar per item) and can be gathered with torch.distributed.all_gather_object by the caller.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

from . import _lib

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
    """Embed an n-item into an nb-item: B -> blockdiag(B, I), vectors zero-extended (the scalar a^T (B + diag c)^-1 d does not
    change). The queue no longer needs it -- the kernels pad in registers -- it documents the identity the kernels rely on."""
    import torch
    if n == nb:
        return a.reshape(-1), B.reshape(-1), c.reshape(-1), d.reshape(-1)
    Bp = torch.eye(nb, dtype=B.dtype, device=B.device)
    Bp[:n, :n] = B.reshape(n, n)

    def z(v):
        out = torch.zeros(nb, dtype=v.dtype, device=v.device)
        out[:n] = v.reshape(n)
        return out
    return z(a), Bp.reshape(-1), z(c), z(d)


class SizeBinnedQueue:
    """matinv_queue_* over torch CUDA tensors. One queue serves one dtype (fixed by the first item)."""

    def __init__(self, bins: Sequence[int] = DEFAULT_BINS, device=None):
        import torch
        self.bins = tuple(sorted(bins))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._q = None
        self._dtype = None
        self._keep = []   # tensors of the pending chunks (the C queue holds raw pointers)
        self._tickets = 0
        self._want_var = True
        self._home_ptr = None  # matinv_queue_stream, once asked for
        self._held = []        # (event, chunks) of flushes issued on it

    def _handle(self, dtype):
        import torch
        if self._q is None:
            self._dtype = dtype
            q = ctypes.c_void_p()
            arr = (ctypes.c_int * len(self.bins))(*self.bins)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().matinv_queue_create(ctypes.byref(q), _lib.F64 if dtype == torch.float64 else _lib.F32,
                                                          arr, len(self.bins)))
            self._q = q
        elif dtype != self._dtype:
            raise TypeError("one queue serves one dtype")
        return self._q

    def home_stream(self, dtype=None):
        """The C queue's own stream as a torch stream (matinv_queue_stream): submitting and flushing inside
        `with torch.cuda.stream(q.home_stream())` keeps a flush on the queue's two hardware queues (include/matinv.h). The queue is
        created here if it does not exist yet (dtype: torch.float32 unless given)."""
        import torch
        q = self._handle(dtype or self._dtype or torch.float32)
        self._home_ptr = int(_lib.lib().matinv_queue_stream(q))
        return torch.cuda.ExternalStream(self._home_ptr, device=self.device)

    def __del__(self):
        try:
            if self._q is not None:
                _lib.lib().matinv_queue_destroy(self._q)
        except Exception:
            pass

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
        ts = [t if t.is_contiguous() else t.contiguous() for t in (As, Bs, Cs, Ds)] + ([Es.contiguous()] if Es is not None else [])
        if any(not t.is_cuda or t.dtype != As.dtype for t in ts):
            raise ValueError("items must be CUDA tensors of one dtype")
        q = self._handle(As.dtype)
        bin_of(n, self.bins)  # raises for n beyond the largest bin, like the C side
        first = ctypes.c_size_t()
        p = [t.data_ptr() for t in ts] + ([None] if Es is None else [])
        rc = _lib.lib().matinv_queue_submit(q, int(n), p[0], p[1], p[2], p[3], p[4], int(count), ctypes.byref(first))
        if rc != _lib.OK:
            raise _lib.MatinvError(rc, _lib.lib().matinv_queue_last_error(q).decode())
        self._keep.append(ts)
        self._tickets += count
        self._want_var = self._want_var and Es is not None
        return int(first.value)

    def chunk_table(self, chunks):
        """Prepare a list of chunks [(n, As, Bs, Cs, Ds[, Es]), ...] for `submit_table`: the pointer arrays the C call takes
        (matinv_queue_submit_chunks). The tensors are kept alive by the table."""
        import torch
        k = len(chunks)
        with_e = all(len(ch) > 5 and ch[5] is not None for ch in chunks)
        ts = [[t.contiguous() for t in ch[1:5]] + ([ch[5].contiguous()] if with_e else []) for ch in chunks]
        dtype = ts[0][0].dtype
        if any(not t.is_cuda or t.dtype != dtype for row in ts for t in row):
            raise ValueError("items must be CUDA tensors of one dtype")
        ns = [int(ch[0]) for ch in chunks]
        counts = [row[0].numel() // n for row, n in zip(ts, ns)]
        for row, n, cnt in zip(ts, ns, counts):
            bin_of(n, self.bins)
            if row[0].numel() != cnt * n or row[1].numel() != cnt * n * n or row[2].numel() != cnt * n or row[3].numel() != cnt * n \
                    or (with_e and row[4].numel() != cnt):
                raise ValueError("inconsistent item shapes")
        arr = lambda j: (ctypes.c_void_p * k)(*[row[j].data_ptr() for row in ts])
        return {"k": k, "n": (ctypes.c_int * k)(*ns), "count": (ctypes.c_size_t * k)(*counts), "ptr": [arr(j) for j in range(4)],
                "e": arr(4) if with_e else None, "tickets": (ctypes.c_size_t * k)(), "keep": ts, "dtype": dtype, "items": sum(counts)}

    def submit_table(self, table):
        """Queue every chunk of a prepared table with ONE C call; returns the first tickets of the chunks."""
        q = self._handle(table["dtype"])
        rc = _lib.lib().matinv_queue_submit_chunks(q, table["k"], table["n"], *table["ptr"], table["e"], table["count"], table["tickets"])
        if rc != _lib.OK:
            raise _lib.MatinvError(rc, _lib.lib().matinv_queue_last_error(q).decode())
        self._keep.append([t for row in table["keep"] for t in row])
        self._tickets += table["items"]
        self._want_var = self._want_var and table["e"] is not None
        return list(table["tickets"])

    def pending(self):
        if self._q is None:
            return {}
        per = (ctypes.c_size_t * len(self.bins))()
        _lib.check(_lib.lib().matinv_queue_pending(self._q, None, per))
        return {b: int(v) for b, v in zip(self.bins, per) if v}

    def flush(self) -> Tuple["torch.Tensor", Optional["torch.Tensor"]]:
        """Run every bin; returns (means, variances or None) in ticket order and empties the queue. Variances are computed
        only when EVERY pending item carried an e (a submission that mixes items with and without e gets means only).
        Asynchronous on torch's current stream. The queue is empty afterwards whether the flush succeeded or raised -- the
        C queue clears itself in both cases, and this mirror follows it."""
        import torch
        total = self._tickets
        if total == 0:
            return torch.empty(0, device=self.device), None
        means = torch.empty(total, dtype=self._dtype, device=self.device)
        variances = torch.empty(total, dtype=self._dtype, device=self.device) if self._want_var else None
        stream = torch.cuda.current_stream(self.device)
        try:
            with torch.cuda.device(self.device):
                rc = _lib.lib().matinv_queue_flush(self._q, ctypes.c_void_p(means.data_ptr()),
                                                   ctypes.c_void_p(variances.data_ptr()) if variances is not None else None,
                                                   ctypes.c_void_p(stream.cuda_stream))
            if rc != _lib.OK:
                raise _lib.MatinvError(rc, _lib.lib().matinv_queue_last_error(self._q).decode())
        finally:
            # the chunks must outlive the asynchronous launches (also those of a flush that failed half way): their memory is
            # returned to torch's caching allocator only after the current stream has passed this point
            if stream.cuda_stream == self._home_ptr:
                # the queue's own stream: it dies with the queue, so no tensor may carry a record_stream() of it (the caching
                # allocator would record an event on the dead stream when such a tensor is freed) -- hold the chunks until an
                # event recorded behind this flush has completed
                ev = torch.cuda.Event()
                ev.record(stream)
                self._held = [(e_, k_) for (e_, k_) in self._held if not e_.query()] + [(ev, self._keep)]
            else:
                for ts in self._keep:
                    for t in ts:
                        t.record_stream(stream)
            self._keep = []
            self._tickets = 0
            self._want_var = True
        return means, variances
