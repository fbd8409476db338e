"""Hash-prefix sharding of one dedup job over the GPUs of a node.

One process per GPU.  Every rank encodes its own reads into fixed-size records
[hash | key words], groups them by owner = (hash >> 40) % world, and ONE all-to-all over
RCCL/xGMI moves each record to its owner, which inserts it into its own disjoint part of
the set.  The keep flags travel back through the reverse all-to-all and are put back into
input order.  First-occurrence-wins holds globally because rank r's reads are the global
indices [r*n, (r+1)*n) of a step and the owner receives records ordered by (source rank,
position at the source).

The device work goes through an `ops` object: HipOps (the C ABI; production) — or whatever
a CPU test injects to exercise the exchange logic under gloo.  There is no CPU fallback in
the product: HipOps raises if the HIP library or a GPU is missing.
"""
from typing import Sequence

import torch


class _DeviceWords:
    """Exposes raw device memory through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr: int, n_words: int):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False),
                                         "version": 3, "strides": None}


class HipOps:
    """The sharding halves of include/fqdupaway.h on one GPU."""

    def __init__(self, engine):
        self.e = engine

    def key_words(self, len0, len1):
        return self.e.key_words(len0, len1)

    def encode(self, segs, n, records):
        self.e.encode_uniform(segs, n, records)

    def partition(self, records, n, key_words, parts, out, counts, origin):
        self.e.partition_records(records, n, key_words, parts, out, counts, origin)

    def recv_buffer(self, n, len0, len1, device):
        """Room for n records at the tail of the engine's key store, as a tensor the all-to-all can
        write into: the records are then inserted where they lie (fqd_reserve_records)."""
        rw = self.key_words(len0, len1) + 1
        ptr = self.e.reserve_records(n, len0, len1)
        return torch.as_tensor(_DeviceWords(ptr, max(1, n * rw)), device=device)

    def insert(self, records, n, len0, len1, keep):
        self.e.insert_records(records, n, len0, len1, keep)

    def scatter(self, flags, origin, n, keep_out):
        self.e.scatter_flags(flags, origin, n, keep_out)

    def sync(self):
        self.e.sync()

    def stream_handle(self):
        return self.e.stream_handle()


class _Round:
    """Buffers of one round in flight (the pipeline keeps two)."""

    def __init__(self, n_max, rw, world, cap_recv, device):
        i64 = torch.int64
        self.records = torch.empty(n_max * rw, dtype=i64, device=device)
        self.grouped = torch.empty(n_max * rw, dtype=i64, device=device)
        self.origin = torch.empty(n_max, dtype=torch.int32, device=device)
        self.counts = torch.zeros(world, dtype=i64, device=device)
        self.recv_counts = torch.zeros(world, dtype=i64, device=device)
        self.keep_recv = torch.empty(cap_recv, dtype=torch.uint8, device=device)
        self.keep_back = torch.empty(n_max, dtype=torch.uint8, device=device)
        self.cap_recv = cap_recv


class ShardedDedup:
    """dedup(segs, n, keep): keep[i] = 1 iff read i of THIS rank's batch is the first with its
    key among all ranks' batches so far (global order: step, then rank, then position)."""

    def __init__(self, ops, dist, device, n_max: int, len0: int, len1: int = 0, slack: float = 1.15):
        self.ops, self.dist, self.device = ops, dist, device
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.len0, self.len1 = len0, len1
        self.W = ops.key_words(len0, len1)
        self.rw = self.W + 1
        self.n_max = n_max
        i64 = torch.int64
        self.records = torch.empty(n_max * self.rw, dtype=i64, device=device)
        self.grouped = torch.empty(n_max * self.rw, dtype=i64, device=device)
        self.origin = torch.empty(n_max, dtype=torch.int32, device=device)
        self.counts = torch.zeros(self.world, dtype=i64, device=device)
        self.recv_counts = torch.zeros(self.world, dtype=i64, device=device)
        self.cap_recv = int(n_max * slack) + 4096
        import os
        # receive straight into the owner's key store (FQD_SHARDED_INPLACE=0: through a staging buffer)
        self.in_place = hasattr(ops, "recv_buffer") and os.environ.get("FQD_SHARDED_INPLACE", "1") != "0"
        self.recv = None if self.in_place else torch.empty(self.cap_recv * self.rw, dtype=i64, device=device)
        self.keep_recv = torch.empty(self.cap_recv, dtype=torch.uint8, device=device)
        self.keep_back = torch.empty(n_max, dtype=torch.uint8, device=device)

    def dedup(self, segs: Sequence, n: int, keep):
        ops, dist, rw = self.ops, self.dist, self.rw
        if n > self.n_max:
            raise ValueError("batch larger than the buffers this ShardedDedup was built for")
        ops.encode(segs, n, self.records)
        ops.partition(self.records, n, self.W, self.world, self.grouped, self.counts, self.origin)
        ops.sync()                                   # the collectives run on torch's stream
        dist.all_to_all_single(self.recv_counts, self.counts)
        send = [int(c) for c in self.counts.tolist()]
        recv = [int(c) for c in self.recv_counts.tolist()]
        n_recv = sum(recv)
        if n_recv > self.cap_recv:                   # a skewed step: grow once, keep going
            self.cap_recv = int(n_recv * 1.1) + 4096
            if not self.in_place:
                self.recv = torch.empty(self.cap_recv * rw, dtype=torch.int64, device=self.device)
            self.keep_recv = torch.empty(self.cap_recv, dtype=torch.uint8, device=self.device)
        recv_buf = ops.recv_buffer(n_recv, self.len0, self.len1, self.device) if self.in_place else self.recv
        dist.all_to_all_single(recv_buf[: n_recv * rw], self.grouped[: n * rw],
                               output_split_sizes=[c * rw for c in recv], input_split_sizes=[c * rw for c in send])
        self._sync_comm()
        ops.insert(recv_buf, n_recv, self.len0, self.len1, self.keep_recv)
        ops.sync()
        dist.all_to_all_single(self.keep_back[:n], self.keep_recv[:n_recv],
                               output_split_sizes=send, input_split_sizes=recv)
        self._sync_comm()
        ops.scatter(self.keep_back, self.origin, n, keep)
        return n_recv

    # ------------------------------------------------------------------------------------
    def dedup_rounds(self, rounds):
        """Runs several rounds [(segs, n, keep), ...] of one step.  On a GPU the rounds are
        software-pipelined: the all-to-all of round k travels over xGMI while round k+1 is
        encoded and partitioned and round k-1 is inserted (HBM-bound), ordered by events between
        the engine's stream and a communication stream — the host only waits for the split
        sizes.  Elsewhere (CPU tests) the rounds simply run one after the other."""
        pipelined = (getattr(self.device, "type", "cpu") == "cuda" and hasattr(self.ops, "stream_handle")
                     and self.in_place and len(rounds) > 1)
        if not pipelined:
            return [self.dedup(segs, n, keep) for segs, n, keep in rounds]
        import os
        if os.environ.get("FQD_SHARDED_PIPELINE", "1") == "0":
            return [self.dedup(segs, n, keep) for segs, n, keep in rounds]
        ops, dist, rw, dev = self.ops, self.dist, self.rw, self.device
        if not hasattr(self, "_pipe"):
            try:
                self._s_e = torch.cuda.ExternalStream(ops.stream_handle(), device=dev)
                self._s_c = torch.cuda.Stream(device=dev)
            except Exception:                                 # no stream interop: keep the simple order
                self.in_place = self.in_place and False
                return [self.dedup(segs, n, keep) for segs, n, keep in rounds]
            self._pipe = [_Round(self.n_max, rw, self.world, self.cap_recv, dev) for _ in range(2)]
        s_e, s_c = self._s_e, self._s_c
        R = len(rounds)
        st = [dict() for _ in range(R)]

        def encode(k):
            segs, n, _ = rounds[k]
            b = self._pipe[k % 2]
            ops.encode(segs, n, b.records)
            ops.partition(b.records, n, self.W, self.world, b.grouped, b.counts, b.origin)
            st[k]["ev_p"] = torch.cuda.Event(); st[k]["ev_p"].record(s_e)

        def exchange(k):
            _, n, _ = rounds[k]
            b = self._pipe[k % 2]
            st[k]["ev_p"].synchronize()                       # host: the split sizes of round k are ready
            with torch.cuda.stream(s_c):                      # reads ordered behind the collective on s_c
                dist.all_to_all_single(b.recv_counts, b.counts)
                send = [int(c) for c in b.counts.tolist()]
                recv = [int(c) for c in b.recv_counts.tolist()]
            n_recv = sum(recv)
            if n_recv > b.cap_recv:
                b.cap_recv = int(n_recv * 1.1) + 4096
                b.keep_recv = torch.empty(b.cap_recv, dtype=torch.uint8, device=dev)
            buf = ops.recv_buffer(n_recv, self.len0, self.len1, dev)     # tail of the key store, after insert(k-1)
            s_c.wait_event(st[k]["ev_p"])
            with torch.cuda.stream(s_c):
                work = dist.all_to_all_single(buf[: n_recv * rw], b.grouped[: n * rw],
                                              output_split_sizes=[c * rw for c in recv],
                                              input_split_sizes=[c * rw for c in send], async_op=True)
            st[k].update(send=send, recv=recv, n_recv=n_recv, buf=buf, work=work)

        def insert(k):
            b = self._pipe[k % 2]
            with torch.cuda.stream(s_c):
                st[k]["work"].wait()
                ev = torch.cuda.Event(); ev.record(s_c)
            s_e.wait_event(ev)                                # the engine's stream waits for the records
            ops.insert(st[k]["buf"], st[k]["n_recv"], self.len0, self.len1, b.keep_recv)
            st[k]["ev_i"] = torch.cuda.Event(); st[k]["ev_i"].record(s_e)

        def give_back(k):
            _, n, keep = rounds[k]
            b = self._pipe[k % 2]
            s_c.wait_event(st[k]["ev_i"])
            with torch.cuda.stream(s_c):
                dist.all_to_all_single(b.keep_back[:n], b.keep_recv[: st[k]["n_recv"]],
                                       output_split_sizes=st[k]["send"], input_split_sizes=st[k]["recv"])
                ev = torch.cuda.Event(); ev.record(s_c)
            s_e.wait_event(ev)
            ops.scatter(b.keep_back, b.origin, n, keep)

        encode(0)
        exchange(0)
        for k in range(R):
            if k + 1 < R:
                encode(k + 1)                                 # runs under the all-to-all of round k
            insert(k)
            if k + 1 < R:
                exchange(k + 1)                               # its all-to-all runs under insert(k)
            give_back(k)
        return [st[k]["n_recv"] for k in range(R)]

    def _sync_comm(self):
        if self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
