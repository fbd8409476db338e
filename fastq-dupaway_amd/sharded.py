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
    """The sharding halves of include/fqdupaway.h on one GPU.  What travels is the key alone (the
    owner recomputes the hash): exchange_words = key words; FQD_SHARDED_WITH_HASH=1 keeps the older
    [hash | key] records on the wire."""

    def __init__(self, engine):
        import os
        self.e = engine
        self.with_hash = os.environ.get("FQD_SHARDED_WITH_HASH") == "1"

    def key_words(self, len0, len1):
        return self.e.key_words(len0, len1)

    def exchange_words(self, len0, len1):
        return self.key_words(len0, len1) + (1 if self.with_hash else 0)

    def encode(self, segs, n, records):
        self.e.encode_uniform(segs, n, records)

    def partition(self, records, n, key_words, parts, out, counts, origin):
        if self.with_hash:
            self.e.partition_records(records, n, key_words, parts, out, counts, origin)
        else:
            self.e.partition_keys(records, n, key_words, parts, out, counts, origin)

    def recv_buffer(self, n, len0, len1, device):
        """Room for n records at the tail of the engine's key store, as a tensor the all-to-all can
        write into: they are then inserted where they lie (fqd_reserve_keys / fqd_reserve_records)."""
        xw = self.exchange_words(len0, len1)
        ptr = self.e.reserve_records(n, len0, len1) if self.with_hash else self.e.reserve_keys(n, len0, len1)
        return torch.as_tensor(_DeviceWords(ptr, max(1, n * xw)), device=device)

    def insert(self, records, n, len0, len1, keep):
        if self.with_hash:
            self.e.insert_records(records, n, len0, len1, keep)
        else:
            self.e.insert_keys(records, n, len0, len1, keep)

    def scatter(self, flags, origin, n, keep_out):
        self.e.scatter_flags(flags, origin, n, keep_out)

    def sync(self):
        self.e.sync()

    def stream_handle(self):
        return self.e.stream_handle()


class _Round:
    """Buffers of one round in flight (the pipeline keeps two)."""

    def __init__(self, n_max, rw, xw, world, cap_recv, device):
        i64 = torch.int64
        self.records = torch.empty(n_max * rw, dtype=i64, device=device)
        self.grouped = torch.empty(n_max * xw, dtype=i64, device=device)
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
        self.rw = self.W + 1                                  # words of an encoded record [hash | key]
        # words of a record on the wire: the device halves may leave the hash out (HipOps)
        self.xw = ops.exchange_words(len0, len1) if hasattr(ops, "exchange_words") else self.rw
        self.n_max = n_max
        i64 = torch.int64
        self.records = torch.empty(n_max * self.rw, dtype=i64, device=device)
        self.grouped = torch.empty(n_max * self.xw, dtype=i64, device=device)
        self.origin = torch.empty(n_max, dtype=torch.int32, device=device)
        self.counts = torch.zeros(self.world, dtype=i64, device=device)
        self.recv_counts = torch.zeros(self.world, dtype=i64, device=device)
        self.cap_recv = int(n_max * slack) + 4096
        import os
        # receive straight into the owner's key store (FQD_SHARDED_INPLACE=0: through a staging buffer)
        self.in_place = hasattr(ops, "recv_buffer") and os.environ.get("FQD_SHARDED_INPLACE", "1") != "0"
        self.recv = None if self.in_place else torch.empty(self.cap_recv * self.xw, dtype=i64, device=device)
        self.keep_recv = torch.empty(self.cap_recv, dtype=torch.uint8, device=device)
        self.keep_back = torch.empty(n_max, dtype=torch.uint8, device=device)

    # No rank-to-rank message of an exchange may exceed this many bytes.  Measured on this image (RCCL
    # 2.26.6, torch 2.10; tools/a2a_probe.py, profiles/r02_a2a_probe.jsonl): of a message above 1 GiB
    # (2^30 bytes exactly is still fine) only the first half arrives — with ordinary torch tensors on
    # both sides just as with the key store aliased in, so it is a limit of the library, not of the
    # buffers used here.  Larger exchanges are cut into passes of equal slices (_all_to_all).
    MAX_MESSAGE = 512 << 20

    def _exchange_counts(self, counts):
        """Every rank's send counts to every rank (world x world, row = sender): the receive counts of this
        rank are a column, and the largest message of the exchange — which decides into how many passes it
        is cut — is known everywhere without another collective."""
        table = torch.empty(self.world * self.world, dtype=torch.int64, device=counts.device)
        self.dist.all_gather_into_tensor(table, counts)
        t = table.view(self.world, self.world).cpu()
        send = [int(c) for c in t[self.rank].tolist()]
        recv = [int(c) for c in t[:, self.rank].tolist()]
        return send, recv, int(t.max().item())

    def _all_to_all(self, out, inp, recv, send, width, largest, async_op=False):
        """all_to_all_single of items `width` elements wide (recv/send in items), cut into passes when the
        largest message anywhere (`largest` items) would exceed MAX_MESSAGE: pass p moves the p-th slice
        of every rank-to-rank message through contiguous staging buffers."""
        dist = self.dist
        item_bytes = width * out.element_size()
        passes = max(1, -(-(largest * item_bytes) // self.MAX_MESSAGE))
        if passes == 1:
            extra = {"async_op": True} if async_op else {}
            return dist.all_to_all_single(out, inp, output_split_sizes=[c * width for c in recv],
                                          input_split_sizes=[c * width for c in send], **extra)
        def cut(c, p):                                   # slice p of a message of c items
            per = -(-c // passes)
            lo = min(c, p * per)
            return lo, min(c, lo + per) - lo
        s_off = [0]; r_off = [0]
        for c in send: s_off.append(s_off[-1] + c)
        for c in recv: r_off.append(r_off[-1] + c)
        for p in range(passes):
            s_cut = [cut(c, p) for c in send]; r_cut = [cut(c, p) for c in recv]
            stage_in = torch.cat([inp[(s_off[j] + lo) * width:(s_off[j] + lo + m) * width] for j, (lo, m) in enumerate(s_cut)])
            stage_out = torch.empty(sum(m for _, m in r_cut) * width, dtype=out.dtype, device=out.device)
            dist.all_to_all_single(stage_out, stage_in, output_split_sizes=[m * width for _, m in r_cut],
                                   input_split_sizes=[m * width for _, m in s_cut])
            at = 0
            for j, (lo, m) in enumerate(r_cut):
                out[(r_off[j] + lo) * width:(r_off[j] + lo + m) * width].copy_(stage_out[at * width:(at + m) * width])
                at += m
        return None

    def dedup(self, segs: Sequence, n: int, keep):
        ops, dist, xw = self.ops, self.dist, self.xw
        if n > self.n_max:
            raise ValueError("batch larger than the buffers this ShardedDedup was built for")
        ops.encode(segs, n, self.records)
        ops.partition(self.records, n, self.W, self.world, self.grouped, self.counts, self.origin)
        ops.sync()                                   # the collectives run on torch's stream
        send, recv, largest = self._exchange_counts(self.counts)
        n_recv = sum(recv)
        if n_recv > self.cap_recv:                   # a skewed step: grow once, keep going
            self.cap_recv = int(n_recv * 1.1) + 4096
            if not self.in_place:
                self.recv = torch.empty(self.cap_recv * xw, dtype=torch.int64, device=self.device)
            self.keep_recv = torch.empty(self.cap_recv, dtype=torch.uint8, device=self.device)
        recv_buf = ops.recv_buffer(n_recv, self.len0, self.len1, self.device) if self.in_place else self.recv
        self._all_to_all(recv_buf[: n_recv * xw], self.grouped[: n * xw], recv, send, xw, largest)
        self._sync_comm()
        ops.insert(recv_buf, n_recv, self.len0, self.len1, self.keep_recv)
        ops.sync()
        self._all_to_all(self.keep_back[:n], self.keep_recv[:n_recv], send, recv, 1, largest)
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
        ops, dist, rw, xw, dev = self.ops, self.dist, self.rw, self.xw, self.device
        if not hasattr(self, "_pipe"):
            try:
                self._s_e = torch.cuda.ExternalStream(ops.stream_handle(), device=dev)
                self._s_c = torch.cuda.Stream(device=dev)
            except Exception:                                 # no stream interop: keep the simple order
                self.in_place = self.in_place and False
                return [self.dedup(segs, n, keep) for segs, n, keep in rounds]
            self._pipe = [_Round(self.n_max, rw, xw, self.world, self.cap_recv, dev) for _ in range(2)]
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
                send, recv, largest = self._exchange_counts(b.counts)
            n_recv = sum(recv)
            if n_recv > b.cap_recv:
                b.cap_recv = int(n_recv * 1.1) + 4096
                b.keep_recv = torch.empty(b.cap_recv, dtype=torch.uint8, device=dev)
            buf = ops.recv_buffer(n_recv, self.len0, self.len1, dev)     # tail of the key store, after insert(k-1)
            s_c.wait_event(st[k]["ev_p"])
            with torch.cuda.stream(s_c):
                work = self._all_to_all(buf[: n_recv * xw], b.grouped[: n * xw], recv, send, xw, largest, async_op=True)
            st[k].update(send=send, recv=recv, n_recv=n_recv, buf=buf, work=work, largest=largest)

        def insert(k):
            b = self._pipe[k % 2]
            with torch.cuda.stream(s_c):
                if st[k]["work"] is not None:
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
                self._all_to_all(b.keep_back[:n], b.keep_recv[: st[k]["n_recv"]], st[k]["send"], st[k]["recv"], 1, st[k]["largest"])
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



class LazyShardedDedup:
    """Optimistic exchange (include/fqdupaway.h, "optimistic sharding"): every rank keeps its
    keys, owners dedup by the 64-bit hash alone, and only candidate duplicates are checked key
    against key at the rank that holds the earlier record.  Per read this moves 16 B (hash +
    payload) out and 8 B (reply) back, plus key + 8 B out and 1 B back per candidate, instead of
    hash + key out and 1 B back for every read.

    `local` is an Engine used as this rank's key store (fqd_encode_batch), `owner` a second Engine
    (segments = 1) used as hash engine.  Results are exact: a candidate whose key differs from
    the earlier record's (two keys, one hash) goes to a small replicated side set that holds
    every such key in global order, so it is still compared with all earlier keys it could equal.
    Global order is (call, rank, position), as in ShardedDedup."""

    MAX_MESSAGE = 256 << 20                                  # bytes per peer in one all-to-all

    def __init__(self, local, owner, dist, device, n_max: int, len0: int, len1: int = 0, slack: float = 1.15):
        self.local, self.owner, self.dist, self.device = local, owner, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.len0, self.len1 = len0, len1
        self.W = local.key_words(len0, len1)
        self.rw = self.W + 1
        self.n_max = n_max
        i64, i32, u8 = torch.int64, torch.int32, torch.uint8
        dev = device
        self.hashes = torch.empty(n_max, dtype=i64, device=dev)
        self.rec16 = torch.empty(2 * n_max, dtype=i64, device=dev)
        self.grouped16 = torch.empty(2 * n_max, dtype=i64, device=dev)
        self.origin = torch.empty(n_max, dtype=i32, device=dev)
        self.counts = torch.zeros(self.world, dtype=i64, device=dev)
        self.recv_counts = torch.zeros(self.world, dtype=i64, device=dev)
        self.reply_back = torch.empty(n_max, dtype=i64, device=dev)
        self.reply_orig = torch.empty(n_max, dtype=i64, device=dev)
        self.req_local = torch.empty(n_max, dtype=i32, device=dev)
        self.verdict_back = torch.empty(n_max, dtype=u8, device=dev)
        self._owner_side(int(n_max * slack) + 4096)
        self.req = self.greq = self.origin2 = self.recv_req = self.verdict = None
        self.local_base = 0                                  # records this rank has encoded so far
        import os
        self.timing = {} if os.environ.get("FQD_LAZY_TIMING") == "1" else None   # host-side ms per phase
        self.side = set()                                    # keys that lost their hash slot to another key
        self.stats = {"reads": 0, "requests": 0, "refuted": 0, "bytes_out": 0}

    def _t(self, name):
        if self.timing is not None:
            import time
            now = time.perf_counter()
            self.timing[name] = self.timing.get(name, 0.0) + (now - self._t0) * 1e3
            self._t0 = now

    def reset(self):
        """Forget everything (the caller resets the two engines)."""
        self.local_base = 0
        self.side.clear()
        self.stats = dict.fromkeys(self.stats, 0)

    def _owner_side(self, cap):
        self.cap_recv = cap
        self.keep_o = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.first_o = torch.empty(cap, dtype=torch.int32, device=self.device)
        self.reply_o = torch.empty(cap, dtype=torch.int64, device=self.device)

    def _request_side(self, cnt):
        if self.req is None or self.req.numel() < cnt * self.rw:
            cap = int(cnt * 1.25) + 1024
            self.req = torch.empty(cap * self.rw, dtype=torch.int64, device=self.device)
            self.greq = torch.empty(cap * self.rw, dtype=torch.int64, device=self.device)
            self.origin2 = torch.empty(cap, dtype=torch.int32, device=self.device)
            self.verdict_grouped = torch.empty(cap, dtype=torch.uint8, device=self.device)

    def _holder_side(self, m):
        if self.recv_req is None or self.recv_req.numel() < m * self.rw:
            cap = int(m * 1.25) + 1024
            self.recv_req = torch.empty(cap * self.rw, dtype=torch.int64, device=self.device)
            self.verdict = torch.empty(cap, dtype=torch.uint8, device=self.device)

    def _sync(self):
        if getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _exchange_counts(self):
        self.dist.all_to_all_single(self.recv_counts, self.counts)
        return [int(c) for c in self.counts.tolist()], [int(c) for c in self.recv_counts.tolist()]

    def dedup(self, segs: Sequence, n: int, keep) -> int:
        """One round: keep[i] (device uint8, n) for this rank's batch.  Returns the number of
        candidate duplicates this rank had checked."""
        L, O, dist, rw = self.local, self.owner, self.dist, self.rw
        if n > self.n_max:
            raise ValueError("batch larger than the buffers this LazyShardedDedup was built for")
        if self.local_base + n >= 1 << 32:
            raise ValueError("more than 2^32 records on one rank")
        base = self.local_base
        if self.timing is not None:
            import time
            self._t0 = time.perf_counter()
        # 1. keys stay here; [hash | (rank, index)] travels to the owner of the hash
        L.encode_batch(segs, n, self.hashes)
        L.make_hash_records(self.hashes, n, (self.rank << 40) | base, self.rec16)
        L.partition_records(self.rec16, n, 1, self.world, self.grouped16, self.counts, self.origin)
        L.sync(); self._t("encode+part16")
        send, recv = self._exchange_counts()
        n_recv = sum(recv)
        if n_recv > self.cap_recv:
            self._owner_side(int(n_recv * 1.1) + 4096)
        buf = torch.as_tensor(_DeviceWords(O.reserve_hashes(n_recv), max(1, 2 * n_recv)), device=self.device)
        dist.all_to_all_single(buf[: 2 * n_recv], self.grouped16[: 2 * n],
                               output_split_sizes=[2 * c for c in recv], input_split_sizes=[2 * c for c in send])
        self._sync(); self._t("a2a hashes")
        # 2. owners dedup by hash; the answer is "first" or where the earlier record lives
        O.insert_hashes(buf, n_recv, self.keep_o, self.first_o)
        O.hash_replies(n_recv, self.keep_o, self.first_o, self.reply_o)
        O.sync(); self._t("insert_hashes+replies")
        dist.all_to_all_single(self.reply_back[:n], self.reply_o[:n_recv], output_split_sizes=send, input_split_sizes=recv)
        self._sync()
        L.scatter_u64(self.reply_back, self.origin, n, self.reply_orig)
        # 3. candidates are checked key against key at the holder of the earlier record
        L.sync(); self._t("a2a replies+scatter")
        self._request_side(max(1024, n // 4) if self.req is None else 0)
        cnt, fits = L.build_requests(self.reply_orig, n, base, self.req, self.req.numel() // rw, self.req_local)
        if not fits:
            self._request_side(cnt)
            cnt, fits = L.build_requests(self.reply_orig, n, base, self.req, self.req.numel() // rw, self.req_local)
        self._t("build_requests")
        self._check_requests(cnt); self._t("check_requests")
        # 4. flags; refuted candidates (rare: two keys, one hash) go through the side set
        refuted = L.apply_replies(self.reply_orig, n, keep, self.verdict_back, cnt)
        self._settle_refuted(cnt, refuted, keep); self._t("apply+settle")
        self.local_base += n
        self.stats["reads"] += n
        self.stats["requests"] += cnt
        self.stats["refuted"] += refuted
        self.stats["bytes_out"] += 16 * n + 8 * n_recv + (8 * rw + 1) * cnt
        return cnt

    def _check_requests(self, cnt):
        """Sends the cnt requests in self.req to the holders, slice by slice so that no message
        exceeds MAX_MESSAGE, and fills verdict_back[k] for request k."""
        L, dist, rw = self.local, self.dist, self.rw
        limit = max(1, self.MAX_MESSAGE // (8 * rw))
        # every rank must run the same number of slices: agree on the largest request count
        t = torch.tensor([cnt], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=_reduce_max(dist))
        most = int(t.item())
        slice_len = max(1, min(most, limit * self.world // 2)) if most else 0
        n_slices = (most + slice_len - 1) // slice_len if most else 0
        for s in range(n_slices):
            lo = min(cnt, s * slice_len)
            m = min(cnt, lo + slice_len) - lo
            self._request_side(max(m, 1))
            req = self.req[lo * rw:]
            L.partition_records(req, m, self.W, self.world, self.greq, self.counts, self.origin2)
            L.sync()
            send, recv = self._exchange_counts()
            if max(send + [0]) * 8 * rw > 4 * self.MAX_MESSAGE:
                raise RuntimeError("request exchange: one holder is the target of too many candidates in a slice")
            m_recv = sum(recv)
            self._holder_side(max(m_recv, 1))
            dist.all_to_all_single(self.recv_req[: m_recv * rw], self.greq[: m * rw],
                                   output_split_sizes=[c * rw for c in recv], input_split_sizes=[c * rw for c in send])
            self._sync()
            L.verify_requests(self.recv_req, m_recv, self.verdict)
            L.sync()
            dist.all_to_all_single(self.verdict_grouped[:m], self.verdict[:m_recv], output_split_sizes=send, input_split_sizes=recv)
            self._sync()
            L.scatter_flags(self.verdict_grouped, self.origin2, m, self.verdict_back[lo:])
        L.sync()

    def _settle_refuted(self, cnt, refuted, keep):
        t = torch.tensor([refuted], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t)
        if int(t.item()) == 0:
            return
        mine = []
        if refuted:
            k = (self.verdict_back[:cnt] == 0).nonzero().flatten()
            rows = self.req.view(-1, self.rw)[k, 1:].cpu().numpy()
            where = self.req_local[k.to(self.req_local.device)].cpu().tolist()
            mine = [(int(i), rows[j].tobytes()) for j, i in enumerate(where)]       # requests are in read order
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine)
        fresh = []
        for r, items in enumerate(everyone):                 # global order: rank, then position
            for i, key in items:
                if key not in self.side:
                    self.side.add(key)
                    if r == self.rank:
                        fresh.append(i)
        if fresh:
            keep[torch.tensor(fresh, dtype=torch.int64, device=keep.device)] = 1
        self._sync()


def _reduce_max(dist):
    op = getattr(dist, "ReduceOp", None)
    return op.MAX if op is not None else torch.distributed.ReduceOp.MAX
