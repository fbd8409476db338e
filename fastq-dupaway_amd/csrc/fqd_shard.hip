// fqd_shard.hip — one dedup job over the GPUs of a node (BASELINE north_star: "reads partitioned
// across the 8 GPUs of one node by hash prefix with an RCCL all-to-all over xGMI so each GPU owns a disjoint
// bucket range").  The reference has nothing to mirror here: it is one thread on one core.
//
// A shard group is `world` ranks, one engine (= one GPU) each; a process hosts n_local consecutive ranks of it:
// all of them (the CLI with FQD_DEVICES, tests with several ranks on one card) or one (bench.py under
// torch.distributed.run, one process per GPU).  Global input order is (round, rank, position).
//
// One round, per rank:
//   encode + group by owner   every read becomes a fixed-size key; owner = (hash >> 40) % world; the keys bound
//                             for owner d are written to SLAB d of the send buffer: `cap` key slots, a few per
//                             cent more than a fair share (fqd_partition_slabs)
//   all-to-all                slab d travels to rank d and lands at slot s*cap of the room rank d reserved at the
//                             tail of its key store — messages of FIXED size, so the exchange is queued before any
//                             count has reached a host; the true counts travel beside the slabs (8 bytes a pair)
//   insert                    the owner inserts its `world` slabs where they lie, in (source rank, position) order,
//                             passing over the unused slots (fqd_insert_slabs): first occurrence wins, globally
//   flags back, scatter       the reverse all-to-all (cap bytes a pair) and keep[origin[slot]] = flag
// Rounds are software-pipelined over two streams per rank: the exchange of round k travels while round k+1 is
// encoded, the insert of round k runs under the exchange of round k+1.  The host looks at a round's counts only to
// decide whether a slab overflowed, and only after the next round's encoder has been queued — it never waits
// with an idle GPU behind it.
//
// A slab that overflows (one owner draws far more than its share: millions of copies of one read) is settled
// before the owner inserts anything of that round: the spilled keys follow in a second, exactly sized exchange —
// both ends of a pair know its true count — and the owner lays the round out again in (source, position) order.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

#include "../../include/fqdupaway.h"
#include "fqd_shard_plan.hpp"

hipStream_t fqd_internal_stream(fqd_engine* e);
int fqd_internal_device(fqd_engine* e);
uint64_t* fqd_internal_state(fqd_engine* e);

namespace {

__global__ __launch_bounds__(256)
void shard_scatter_flags_kernel(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ origin, uint64_t n,
                                uint8_t* __restrict__ keep_out)
{
    for (uint64_t k = blockIdx.x * uint64_t(256) + threadIdx.x; k < n; k += uint64_t(gridDim.x) * 256) {
        const uint32_t to = origin[k];
        if (to != 0xFFFFFFFFu) keep_out[to] = flags[k];
    }
}

struct Round {                           // buffers and events of one round in flight (two per rank)
    uint64_t* grouped = nullptr;         // world slabs of cap keys, then the overflow region (round_reads keys)
    uint32_t* origin = nullptr;          // input position per slot of `grouped`
    uint8_t*  keep_recv = nullptr;       // owner side: flags of what was inserted
    uint8_t*  keep_back = nullptr;       // source side: flags per slot of `grouped`
    uint64_t* d_counts = nullptr;        // device: [world] out (true counts per owner) then [world] in (per source)
    uint64_t* h_counts = nullptr;        // pinned mirror
    uint64_t* slot = nullptr;            // where this round's keys are received (tail of the key store)
    hipEvent_t ev_part = nullptr, ev_xchg = nullptr, ev_spill = nullptr, ev_ins = nullptr, ev_done = nullptr, t0 = nullptr, t1 = nullptr;
    uint64_t n = 0, n_inserted = 0;
    fqd_reads seg[2] = {};               // the round's input (it must stay where it is until the round's flags are final)
    uint8_t* keep_dst = nullptr;
    bool used = false;
    bool compact = false;                // owner side: a spill arrived and the round was laid out again, exactly
    uint64_t* h_bad = nullptr;           // pinned: the engine's first-bad-byte word once this round was encoded
};

struct Local {
    fqd_engine* e = nullptr; int device = 0; int rank = 0;
    hipStream_t es = nullptr, cs = nullptr;
    ncclComm_t comm = nullptr;
    uint64_t* records = nullptr;
    uint64_t* spill = nullptr; size_t spill_cap = 0;     // rare path: a round laid out again
    Round rb[2];
    fqd_shard_stats st{};
};

} // namespace

struct fqd_shard {
    fqd_shard_config cfg{};
    std::vector<Local> lr;
    uint32_t S = 1, K = 0;
    bool padded = false; uint32_t own_len0 = 0, own_len1 = 0;
    uint64_t cap = 0;
    uint64_t rounds = 0;                 // rounds started
    int64_t  pending = -1;               // round whose receive side has not been finished yet
    bool use_rccl = false;
    std::string err;
    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {

thread_local std::string g_shard_error;

#define SH_HIP(s, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); \
    return (s)->fail(FQD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
#define SH_NCCL(s, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return (s)->fail(FQD_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); } while (0)
#define SH_ENG(s, l, expr) do { int rc_ = (expr); if (rc_ != FQD_OK) \
    return (s)->fail(rc_, std::string("rank ") + std::to_string((l).rank) + ": " + fqd_last_error((l).e)); } while (0)

constexpr size_t kPiece = size_t(512) << 20;     // RCCL 2.26.6 on this image loses the second half of a message above 1 GiB (tools/a2a_probe.py)

// One byte range from a buffer of one rank to a buffer of another.
struct Xfer { int src, dst; const void* from; void* to; size_t bytes; };

int local_of(const fqd_shard* s, int rank) { const int l = rank - s->cfg.first_rank; return (l >= 0 && l < s->cfg.n_local) ? l : -1; }

// Moves the transfers on the ranks' communication streams.  RCCL: every transfer whose source or destination is a
// local rank becomes a send and/or a receive inside ONE group (a direct all-to-all over the links, not a ring).
// Copies (every rank in this process): one asynchronous copy per transfer on the destination's stream, behind the
// event `ready[src]` that says the source bytes exist.
int move(fqd_shard* s, const std::vector<Xfer>& xs, const std::vector<hipEvent_t>& ready)
{
    if (s->use_rccl) {
        for (Local& l : s->lr) { SH_HIP(s, hipSetDevice(l.device)); SH_HIP(s, hipStreamWaitEvent(l.cs, ready[size_t(l.rank - s->cfg.first_rank)], 0)); }
        SH_NCCL(s, ncclGroupStart());
        for (const Xfer& x : xs)
            for (size_t at = 0; at < x.bytes; at += kPiece) {
                const size_t m = std::min(kPiece, x.bytes - at);
                const int ls = local_of(s, x.src), ld = local_of(s, x.dst);
                if (ls >= 0) SH_NCCL(s, ncclSend(static_cast<const char*>(x.from) + at, m, ncclUint8, x.dst, s->lr[size_t(ls)].comm, s->lr[size_t(ls)].cs));
                if (ld >= 0) SH_NCCL(s, ncclRecv(static_cast<char*>(x.to) + at, m, ncclUint8, x.src, s->lr[size_t(ld)].comm, s->lr[size_t(ld)].cs));
            }
        SH_NCCL(s, ncclGroupEnd());
        for (const Xfer& x : xs) {
            const int ls = local_of(s, x.src), ld = local_of(s, x.dst);
            if (ls >= 0) s->lr[size_t(ls)].st.bytes_sent += x.bytes;
            if (ld >= 0) s->lr[size_t(ld)].st.bytes_received += x.bytes;
        }
        return FQD_OK;
    }
    // every destination stream first waits for every source it reads from
    for (Local& d : s->lr) {
        SH_HIP(s, hipSetDevice(d.device));
        for (Local& src : s->lr) SH_HIP(s, hipStreamWaitEvent(d.cs, ready[size_t(src.rank - s->cfg.first_rank)], 0));
    }
    for (const Xfer& x : xs) {
        Local& d = s->lr[size_t(local_of(s, x.dst))]; Local& f = s->lr[size_t(local_of(s, x.src))];
        SH_HIP(s, hipSetDevice(d.device));
        if (d.device == f.device) SH_HIP(s, hipMemcpyAsync(x.to, x.from, x.bytes, hipMemcpyDeviceToDevice, d.cs));
        else                      SH_HIP(s, hipMemcpyPeerAsync(x.to, d.device, x.from, f.device, x.bytes, d.cs));
        f.st.bytes_sent += x.bytes; d.st.bytes_received += x.bytes;
    }
    return FQD_OK;
}

// In a multi-process group a rank only knows its own buffers: a transfer's far end is then described by what
// both ends can compute (slab geometry), and `from`/`to` of the far side stay null — move() never touches them.
const void* at_words(const uint64_t* p, uint64_t words) { return p ? p + words : nullptr; }

Round* round_of(fqd_shard* s, int rank, uint64_t k) { const int l = local_of(s, rank); return l >= 0 ? &s->lr[size_t(l)].rb[k & 1] : nullptr; }

// Queues round k's all-to-all: slabs out, counts beside them.
int exchange_forward(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const uint64_t slab_words = s->cap * s->K;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        SH_ENG(s, l, fqd_reserve_keys(l.e, uint64_t(W) * s->cap, s->own_len0, s->own_len1, &r.slot));
        SH_HIP(s, hipEventRecord(r.t0, l.cs));
    }
    std::vector<Xfer> xs;
    std::vector<hipEvent_t> ready;
    for (Local& l : s->lr) ready.push_back(l.rb[k & 1].ev_part);
    for (int src = 0; src < W; ++src)
        for (int dst = 0; dst < W; ++dst) {
            Round* a = round_of(s, src, k); Round* b = round_of(s, dst, k);
            if (!a && !b) continue;
            xs.push_back({src, dst, a ? at_words(a->grouped, fqd_plan::slab_slot(uint32_t(dst), s->cap) * s->K) : nullptr,
                          b ? const_cast<void*>(at_words(b->slot, fqd_plan::slab_slot(uint32_t(src), s->cap) * s->K)) : nullptr, slab_words * 8});
            xs.push_back({src, dst, a ? at_words(a->d_counts, uint64_t(dst)) : nullptr,
                          b ? const_cast<void*>(at_words(b->d_counts, uint64_t(W) + uint64_t(src))) : nullptr, 8});
        }
    int rc = move(s, xs, ready);
    if (rc) return rc;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipEventRecord(r.t1, l.cs));
        SH_HIP(s, hipMemcpyAsync(r.h_counts + W, r.d_counts + W, size_t(W) * 8, hipMemcpyDeviceToHost, l.cs));
        SH_HIP(s, hipEventRecord(r.ev_xchg, l.cs));
        l.st.slab_records = s->cap;
    }
    return FQD_OK;
}

// The owner side of round k once its keys have arrived: insert (after settling any overflow).
int finish_receive(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const uint64_t cap = s->cap, K = s->K;
    // the host reads the counts here — long after the exchange was queued, with the next encoder already behind it
    bool any_over = false;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipEventSynchronize(r.ev_xchg));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.t0, r.t1) == hipSuccess) l.st.exchange_ms += ms;
        r.compact = fqd_plan::owner_is_compact(r.h_counts + W, uint32_t(W), cap);
        bool over = r.compact;
        for (int p = 0; p < W; ++p) if (r.h_counts[p] > cap) over = true;
        if (over) l.st.overflow_rounds++;
        any_over = any_over || over;
    }
    if (any_over) {
        // ---- a slab overflowed: the spilled keys follow, exactly sized; both ends of a pair know its count ----
        std::vector<Xfer> xs;
        std::vector<hipEvent_t> ready;
        for (Local& l : s->lr) {
            Round& r = l.rb[k & 1];
            bool over_out = false;
            for (int p = 0; p < W; ++p) over_out = over_out || r.h_counts[p] > cap;
            if (over_out && !s->padded) {
                // the one-pass encoder leaves out what a full slab cannot take: this source groups the round again, the
                // three-step way, which writes the spill region (same slabs, same origin: nothing already sent changes)
                SH_HIP(s, hipSetDevice(l.device));
                SH_ENG(s, l, fqd_encode_slabs(l.e, r.seg, r.n, uint32_t(W), cap, r.grouped, r.d_counts, r.origin, FQD_SLABS_EXACT));
                SH_HIP(s, hipEventRecord(r.ev_part, l.es));
            }
            ready.push_back(r.ev_part);                        // the sources' send buffers are complete behind this
            if (!r.compact) continue;
            SH_HIP(s, hipSetDevice(l.device));
            // owner side: [all slabs as received][spill of source 0][of source 1]... in a buffer of its own
            const size_t need = fqd_plan::spill_slot(r.h_counts + W, uint32_t(W), uint32_t(W), cap) * K * 8;
            if (need > l.spill_cap) {
                SH_HIP(s, hipStreamSynchronize(l.cs)); SH_HIP(s, hipStreamSynchronize(l.es));
                if (l.spill) SH_HIP(s, hipFree(l.spill));
                l.spill = nullptr; l.spill_cap = 0;
                SH_HIP(s, hipMalloc(reinterpret_cast<void**>(&l.spill), need));
                l.spill_cap = need;
            }
            SH_HIP(s, hipMemcpyAsync(l.spill, r.slot, uint64_t(W) * cap * K * 8, hipMemcpyDeviceToDevice, l.cs));
        }
        for (int src = 0; src < W; ++src)
            for (int dst = 0; dst < W; ++dst) {
                Round* a = round_of(s, src, k); Round* b = round_of(s, dst, k);
                if (!a && !b) continue;
                const uint64_t c = a ? a->h_counts[dst] : b->h_counts[W + src];
                if (c <= cap) continue;
                const void* from = nullptr; void* to = nullptr;
                if (a) from = a->grouped + fqd_plan::spill_slot(a->h_counts, uint32_t(W), uint32_t(dst), cap) * K;
                if (b) to = s->lr[size_t(local_of(s, dst))].spill + fqd_plan::spill_slot(b->h_counts + W, uint32_t(W), uint32_t(src), cap) * K;
                xs.push_back({src, dst, from, to, (c - cap) * K * 8});
            }
        if (!xs.empty()) { const int rc = move(s, xs, ready); if (rc) return rc; }
        for (Local& l : s->lr) { SH_HIP(s, hipSetDevice(l.device)); SH_HIP(s, hipEventRecord(l.rb[k & 1].ev_spill, l.cs)); }
    }
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        if (!r.compact) {
            SH_HIP(s, hipStreamWaitEvent(l.es, r.ev_xchg, 0));
            SH_ENG(s, l, fqd_insert_slabs(l.e, r.slot, uint32_t(W), cap, r.d_counts + W, s->own_len0, s->own_len1, r.keep_recv));
            r.n_inserted = uint64_t(W) * cap;
        } else {
            // this owner received a spill: the round is laid out again, exactly, in (source, position) order
            const uint64_t total = fqd_plan::owner_records(r.h_counts + W, uint32_t(W), cap);
            SH_HIP(s, hipStreamWaitEvent(l.es, r.ev_spill, 0));
            uint64_t* slot = nullptr;
            SH_ENG(s, l, fqd_reserve_keys(l.e, total, s->own_len0, s->own_len1, &slot));
            uint64_t at = 0, spill_at = uint64_t(W) * cap;
            for (int p = 0; p < W; ++p) {
                const uint64_t c = r.h_counts[W + p], head = std::min(c, cap);
                if (head) SH_HIP(s, hipMemcpyAsync(slot + at * K, l.spill + uint64_t(p) * cap * K, head * K * 8, hipMemcpyDeviceToDevice, l.es));
                at += head;
                if (c > cap) {
                    SH_HIP(s, hipMemcpyAsync(slot + at * K, l.spill + spill_at * K, (c - cap) * K * 8, hipMemcpyDeviceToDevice, l.es));
                    at += c - cap; spill_at += c - cap;
                }
            }
            if (total) SH_ENG(s, l, fqd_insert_keys(l.e, slot, total, s->own_len0, s->own_len1, r.keep_recv));
            r.n_inserted = total;
        }
        SH_HIP(s, hipEventRecord(r.ev_ins, l.es));
    }
    return FQD_OK;
}

// Flags of round k back to where the reads came from, and into input order.
int return_flags(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const uint64_t cap = s->cap;
    std::vector<Xfer> xs;
    std::vector<hipEvent_t> ready;
    for (Local& l : s->lr) ready.push_back(l.rb[k & 1].ev_ins);
    for (int own = 0; own < W; ++own)
        for (int src = 0; src < W; ++src) {
            Round* o = round_of(s, own, k); Round* a = round_of(s, src, k);
            if (!o && !a) continue;
            // where source `src`'s flags start at the owner: slot src*cap in the slab layout, the running sum of the
            // true counts in a round laid out again — only the owner needs to know which
            const uint64_t c = a ? a->h_counts[own] : o->h_counts[W + src];
            const uint8_t* from = o ? o->keep_recv + fqd_plan::owner_offset(o->h_counts + W, uint32_t(W), uint32_t(src), cap) : nullptr;
            xs.push_back({own, src, from, a ? a->keep_back + fqd_plan::slab_slot(uint32_t(own), cap) : nullptr, cap});   // fixed size; a short slab's tail means nothing
            if (c > cap)
                xs.push_back({own, src, o ? from + cap : nullptr,
                              a ? a->keep_back + fqd_plan::spill_slot(a->h_counts, uint32_t(W), uint32_t(own), cap) : nullptr, c - cap});
        }
    int rc = move(s, xs, ready);
    if (rc) return rc;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        const uint64_t slots = fqd_plan::spill_slot(r.h_counts, uint32_t(W), uint32_t(W), cap);
        const uint32_t grid = uint32_t(std::min<uint64_t>((slots + 255) / 256, 2048));
        hipLaunchKernelGGL(shard_scatter_flags_kernel, dim3(grid), dim3(256), 0, l.cs,
                           static_cast<const uint8_t*>(r.keep_back), static_cast<const uint32_t*>(r.origin), slots, r.keep_dst);
        SH_HIP(s, hipGetLastError());
        SH_HIP(s, hipEventRecord(r.ev_done, l.cs));
    }
    return FQD_OK;
}

int finish_pending(fqd_shard* s)
{
    if (s->pending < 0) return FQD_OK;
    const uint64_t k = uint64_t(s->pending);
    int rc = finish_receive(s, k);
    if (rc) return rc;
    rc = return_flags(s, k);
    if (rc) return rc;
    s->pending = -1;
    return FQD_OK;
}

void free_all(fqd_shard* s)
{
    for (Local& l : s->lr) {
        (void)hipSetDevice(l.device);
        if (l.cs) (void)hipStreamSynchronize(l.cs);
        if (l.es) (void)hipStreamSynchronize(l.es);
        if (l.comm) (void)ncclCommDestroy(l.comm);
        for (Round& r : l.rb) {
            if (r.grouped) (void)hipFree(r.grouped);
            if (r.origin) (void)hipFree(r.origin);
            if (r.keep_recv) (void)hipFree(r.keep_recv);
            if (r.keep_back) (void)hipFree(r.keep_back);
            if (r.d_counts) (void)hipFree(r.d_counts);
            if (r.h_counts) (void)hipHostFree(r.h_counts);
            if (r.h_bad) (void)hipHostFree(r.h_bad);
            for (hipEvent_t ev : {r.ev_part, r.ev_xchg, r.ev_spill, r.ev_ins, r.ev_done, r.t0, r.t1}) if (ev) (void)hipEventDestroy(ev);
        }
        if (l.records) (void)hipFree(l.records);
        if (l.spill) (void)hipFree(l.spill);
        if (l.cs) (void)hipStreamDestroy(l.cs);
    }
}

} // namespace

extern "C" {

const char* fqd_shard_last_error(const fqd_shard* s) { return s ? s->err.c_str() : g_shard_error.c_str(); }

int fqd_shard_unique_id(uint8_t* id)
{
    if (!id) return FQD_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) <= FQD_SHARD_ID_BYTES, "FQD_SHARD_ID_BYTES too small");
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) { g_shard_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return FQD_ERR_HIP; }
    std::memset(id, 0, FQD_SHARD_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return FQD_OK;
}

uint64_t fqd_shard_slab_capacity(uint64_t round_reads, int32_t world, uint32_t slack_permille)
{
    if (world <= 0) return 0;
    const uint64_t fair = (round_reads + uint64_t(world) - 1) / uint64_t(world);
    const uint64_t slack = slack_permille ? slack_permille : 30u;
    // a fair share plus slack plus six standard deviations of a binomial share, rounded to whole 16-key units
    uint64_t sd6 = 0; while ((sd6 + 1) * (sd6 + 1) <= fair) ++sd6; sd6 *= 6;
    return world == 1 ? std::max<uint64_t>(round_reads, 16) : ((fair + fair * slack / 1000 + sd6 + 64 + 15) & ~15ull);
}

int fqd_shard_create(fqd_engine* const* engines, const fqd_shard_config* cfg, fqd_shard** out)
{
    if (!engines || !cfg || !out || cfg->world <= 0 || cfg->n_local <= 0 || cfg->first_rank < 0 ||
        cfg->first_rank + cfg->n_local > cfg->world || cfg->round_reads == 0 || cfg->len0 == 0) {
        g_shard_error = "fqd_shard_create: bad arguments"; return FQD_ERR_ARG;
    }
    *out = nullptr;
    fqd_shard* s = new fqd_shard();
    s->cfg = *cfg;
    s->padded = (cfg->flags & FQD_SHARD_PADDED) != 0;
    s->K = s->padded ? fqd_padded_key_words(cfg->len0, cfg->len1) : fqd_key_words(cfg->len0, cfg->len1);
    s->S = cfg->len1 ? 2u : 1u;
    s->own_len0 = s->padded ? s->K : cfg->len0;                   // what the owners' engines are told their keys are
    s->own_len1 = s->padded ? FQD_OPAQUE_KEYS : cfg->len1;
    s->cap = fqd_shard_slab_capacity(cfg->round_reads, cfg->world, cfg->slack_permille);
    if (cfg->slab_records) s->cap = cfg->slab_records;
    bool distinct = true;
    for (int a = 0; a < cfg->n_local; ++a) {
        if (!engines[a]) { g_shard_error = "fqd_shard_create: null engine"; delete s; return FQD_ERR_ARG; }
        for (int b = 0; b < a; ++b) if (fqd_internal_device(engines[a]) == fqd_internal_device(engines[b])) distinct = false;
    }
    s->use_rccl = cfg->transport == FQD_SHARD_RCCL;
    if (s->use_rccl && !distinct) s->use_rccl = false;                 // ranks that share a GPU (rehearsals): RCCL refuses them
    if (!s->use_rccl && cfg->n_local != cfg->world) {
        g_shard_error = "fqd_shard_create: peer copies need every rank in this process"; delete s; return FQD_ERR_ARG;
    }
    if (s->use_rccl && !cfg->unique_id) { g_shard_error = "fqd_shard_create: RCCL needs the group's unique id"; delete s; return FQD_ERR_ARG; }
    auto bail = [&](const std::string& m) { g_shard_error = m; free_all(s); delete s; return FQD_ERR_HIP; };
    const uint64_t W = uint64_t(cfg->world), slots = W * s->cap + cfg->round_reads;
    s->lr.resize(size_t(cfg->n_local));
    for (int a = 0; a < cfg->n_local; ++a) {
        Local& l = s->lr[size_t(a)];
        l.e = engines[a]; l.device = fqd_internal_device(l.e); l.rank = cfg->first_rank + a; l.es = fqd_internal_stream(l.e);
        hipError_t err;
        if ((err = hipSetDevice(l.device)) != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(err));
        if ((err = hipStreamCreateWithFlags(&l.cs, hipStreamNonBlocking)) != hipSuccess) return bail(std::string("hipStreamCreate: ") + hipGetErrorString(err));
        if (s->padded && (err = hipMalloc(reinterpret_cast<void**>(&l.records), cfg->round_reads * uint64_t(s->K + 1) * 8)) != hipSuccess)
            return bail(std::string("hipMalloc(records): ") + hipGetErrorString(err));
        for (Round& r : l.rb) {
            if ((err = hipMalloc(reinterpret_cast<void**>(&r.grouped), slots * s->K * 8)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.origin), slots * 4)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.keep_recv), W * s->cap + W * cfg->round_reads + s->cap)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.keep_back), slots + s->cap)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.d_counts), 2 * W * 8)) != hipSuccess ||
                (err = hipHostMalloc(reinterpret_cast<void**>(&r.h_counts), 2 * W * 8, hipHostMallocDefault)) != hipSuccess ||
                (err = hipHostMalloc(reinterpret_cast<void**>(&r.h_bad), 8, hipHostMallocDefault)) != hipSuccess)
                return bail(std::string("shard buffers: ") + hipGetErrorString(err));
            for (hipEvent_t* ev : {&r.ev_part, &r.ev_xchg, &r.ev_spill, &r.ev_ins, &r.ev_done})
                if ((err = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(std::string("hipEventCreate: ") + hipGetErrorString(err));
            for (hipEvent_t* ev : {&r.t0, &r.t1})
                if ((err = hipEventCreate(ev)) != hipSuccess) return bail(std::string("hipEventCreate: ") + hipGetErrorString(err));
        }
        l.st.transport = s->use_rccl ? FQD_SHARD_RCCL : FQD_SHARD_COPY;
    }
    if (s->use_rccl) {
        ncclUniqueId u;
        std::memcpy(&u, cfg->unique_id, sizeof u);
        // RCCL announces itself on stdout when NCCL_DEBUG asks for it; stdout belongs to the CLI's -v lines
        std::fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        ncclResult_t r = ncclGroupStart();
        for (Local& l : s->lr) {
            if (r != ncclSuccess) break;
            (void)hipSetDevice(l.device);
            r = ncclCommInitRank(&l.comm, cfg->world, u, l.rank);
        }
        const ncclResult_t r2 = ncclGroupEnd();
        std::fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); (void)close(saved); }
        if (r != ncclSuccess || r2 != ncclSuccess) return bail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
        for (Local& l : s->lr) { int n = 0; if (ncclCommCount(l.comm, &n) == ncclSuccess) l.st.ranks_in_comm = n; }
    } else {
        for (Local& a : s->lr)
            for (Local& b : s->lr)
                if (a.device != b.device) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) { (void)hipSetDevice(a.device); (void)hipDeviceEnablePeerAccess(b.device, 0); (void)hipGetLastError(); }
                }
        for (Local& l : s->lr) l.st.ranks_in_comm = cfg->world;
    }
    *out = s;
    return FQD_OK;
}

int fqd_shard_destroy(fqd_shard* s)
{
    if (!s) return FQD_OK;
    free_all(s);
    delete s;
    return FQD_OK;
}

int fqd_shard_round(fqd_shard* s, const fqd_reads* seg, const uint64_t* n, uint8_t* const* keep)
{
    if (!s) return FQD_ERR_ARG;
    if (!seg || !n || !keep) return s->fail(FQD_ERR_ARG, "fqd_shard_round: bad arguments");
    const uint64_t k = s->rounds;
    const int W = s->cfg.world;
    for (size_t a = 0; a < s->lr.size(); ++a) {
        Local& l = s->lr[a]; Round& r = l.rb[k & 1];
        if (n[a] > s->cfg.round_reads) return s->fail(FQD_ERR_ARG, "fqd_shard_round: more reads than the group's round size");
        if (n[a] && !keep[a]) return s->fail(FQD_ERR_ARG, "fqd_shard_round: null keep");
        for (uint32_t m = 0; m < s->S; ++m) {
            const fqd_reads& x = seg[a * s->S + m];
            if (n[a] && !s->padded && (x.offsets || x.lengths || x.uniform_len != (m ? s->cfg.len1 : s->cfg.len0)))
                return s->fail(FQD_ERR_ARG, "fqd_shard_round: reads of the group's fixed length(s), equally spaced, are expected (FQD_SHARD_PADDED takes any)");
            if (n[a] && s->padded && !x.lengths && x.uniform_len > (m ? s->cfg.len1 : s->cfg.len0))
                return s->fail(FQD_ERR_ARG, "fqd_shard_round: reads longer than the group's maximum");
        }
        SH_HIP(s, hipSetDevice(l.device));
        if (r.used) SH_HIP(s, hipStreamWaitEvent(l.es, r.ev_done, 0));     // round k-2 has left these buffers
        r.used = true; r.n = n[a]; r.keep_dst = keep[a]; r.compact = false;
        for (uint32_t m = 0; m < s->S; ++m) r.seg[m] = seg[a * s->S + m];
        if (s->padded) {
            SH_ENG(s, l, fqd_encode_padded(l.e, seg + a * s->S, n[a], s->cfg.len0, s->cfg.len1, l.records));
            SH_ENG(s, l, fqd_partition_slabs(l.e, l.records, n[a], s->K, uint32_t(W), s->cap, r.grouped, r.d_counts, r.origin));
        } else {
            // one pass where it applies; a slab that overflows is noticed from the counts and the round grouped again (finish_receive)
            SH_ENG(s, l, fqd_encode_slabs(l.e, seg + a * s->S, n[a], uint32_t(W), s->cap, r.grouped, r.d_counts, r.origin, 0u));
        }
        SH_HIP(s, hipMemcpyAsync(r.h_counts, r.d_counts, size_t(W) * 8, hipMemcpyDeviceToHost, l.es));
        SH_HIP(s, hipMemcpyAsync(r.h_bad, fqd_internal_state(l.e), 8, hipMemcpyDeviceToHost, l.es));     // first bad byte so far, this round's encoder included
        SH_HIP(s, hipEventRecord(r.ev_part, l.es));
        l.st.rounds++;
    }
    int rc;
    // the previous round's owner side: its keys arrived while this round was being queued
    const int64_t prev = s->pending;
    if (prev >= 0 && (rc = finish_receive(s, uint64_t(prev)))) return rc;
    // this round's own counts are needed on the host only to size a spill, one round from now: the exchange is queued blind
    if ((rc = exchange_forward(s, k))) return rc;
    if (prev >= 0 && (rc = return_flags(s, uint64_t(prev)))) return rc;
    s->pending = int64_t(k);
    s->rounds = k + 1;
    return FQD_OK;
}

int fqd_shard_flush(fqd_shard* s)
{
    if (!s) return FQD_ERR_ARG;
    int rc = finish_pending(s);
    if (rc) return rc;
    int first_bad = FQD_OK;
    for (Local& l : s->lr) {
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipStreamSynchronize(l.cs));
        const int erc = fqd_engine_sync(l.e);
        if (erc == FQD_ERR_BAD_BASE) { if (first_bad == FQD_OK) { first_bad = erc; s->err = std::string("rank ") + std::to_string(l.rank) + ": " + fqd_last_error(l.e); } }
        else if (erc != FQD_OK) return s->fail(erc, std::string("rank ") + std::to_string(l.rank) + ": " + fqd_last_error(l.e));
    }
    for (Local& l : s->lr) { SH_HIP(s, hipSetDevice(l.device)); SH_HIP(s, hipStreamSynchronize(l.cs)); }
    return first_bad;
}

int fqd_shard_wait(fqd_shard* s, uint64_t round)
{
    if (!s) return FQD_ERR_ARG;
    if (round >= s->rounds || (s->pending >= 0 && round >= uint64_t(s->pending)) || round + 3 < s->rounds)
        return s->fail(FQD_ERR_ARG, "fqd_shard_wait: that round's flags are not on their way (they follow the next round or a flush) or its buffers were reused");
    bool bad = false;
    for (Local& l : s->lr) {
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipEventSynchronize(l.rb[round & 1].ev_done));
        if (round + 2 >= s->rounds && l.rb[round & 1].h_bad[0] != ~0ull) bad = true;     // (h_bad belongs to round+2 once that one has been started)
    }
    return bad ? s->fail(FQD_ERR_BAD_BASE, "a byte outside {A,C,G,T,N}: see fqd_shard_bad_base") : FQD_OK;
}

int fqd_shard_bad_base(fqd_shard* s, uint64_t round, int32_t* local_rank, uint64_t* record, uint32_t* segment, uint32_t* position, uint8_t* byte)
{
    if (!s || round >= s->rounds || round + 2 < s->rounds) return FQD_ERR_ARG;
    for (size_t a = 0; a < s->lr.size(); ++a) {                 // global order within a round is (rank, position)
        const uint64_t w = s->lr[a].rb[round & 1].h_bad[0];
        if (w == ~0ull) continue;
        if (local_rank) *local_rank = int32_t(a);
        if (record) *record = w >> 32;
        if (segment) *segment = uint32_t((w >> 31) & 1u);
        if (position) *position = uint32_t((w >> 8) & 0x7FFFFFu);
        if (byte) *byte = uint8_t(w & 0xFFu);
        return FQD_OK;
    }
    return FQD_ERR_ARG;
}

int fqd_shard_get_stats(fqd_shard* s, int32_t local_rank, fqd_shard_stats* out)
{
    if (!s || !out || local_rank < 0 || size_t(local_rank) >= s->lr.size()) return FQD_ERR_ARG;
    *out = s->lr[size_t(local_rank)].st;
    out->slab_records = s->cap;
    return FQD_OK;
}

} // extern "C"
