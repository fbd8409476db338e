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
//                             for owner d are written to SLAB d of the send buffer.  The batch is cut into chunks
//                             of consecutive reads and every slab into one SUB-SLAB per chunk (a fair share of a
//                             chunk + 5.5 sigma): that is what lets the encoder write every key ONCE, straight
//                             to its place, in input order, no workgroup waiting for another (fqd_encode_slabs)
//   all-to-all                slab d travels to rank d and lands at slot s*cap of the room rank d reserved at the
//                             tail of its key store — messages of FIXED size, so the exchange is queued before any
//                             count has reached a host; the true counts travel beside the slabs (8 bytes a pair)
//   insert                    the owner inserts its `world` slabs (world * chunks sub-slabs) where they lie, in
//                             (source rank, position) order, passing over the unused slots (fqd_insert_slabs):
//                             first occurrence wins, globally
//   flags back, scatter       the reverse all-to-all (cap bytes a pair) and keep[origin[slot]] = flag
// Rounds are software-pipelined over two streams per rank: the exchange of round k travels while round k+1 is
// encoded, the insert of round k runs under the exchange of round k+1.  The host looks at a round's counts only to
// decide whether a slab overflowed, and only after the next round's encoder has been queued — it never waits
// with an idle GPU behind it.
//
// What does not fit is settled before the owner inserts anything of that round, by a second exchange whose sizes both
// ends of a pair know from the counts of the first: a source one of whose sub-slabs overflowed (the one-pass encoder
// leaves such keys out) groups the round again the three-step way and sends the slab again, filled from its first slot
// on; what a whole slab cannot take (one owner drawing far more than its share: millions of copies of one read) follows
// exactly sized, and the owner lays the round out again with the spills as further sub-slabs behind their slabs.
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

using fqd_plan::Geometry;

// A round's counts, device and pinned mirror alike, as uint64 words:
//   [out chunk counts: W*G][out totals: W][out layout: 1 = 1 when the slabs are filled from their first slot on]
//   [in chunk counts: W*G][in totals: W][in layout: W]
struct CountLayout {
    uint64_t W, G;
    uint64_t oc() const { return 0; }
    uint64_t ot() const { return W * G; }
    uint64_t of() const { return W * G + W; }
    uint64_t ic() const { return W * G + W + 1; }
    uint64_t it() const { return ic() + W * G; }
    uint64_t il() const { return it() + W; }
    uint64_t words() const { return il() + W; }
};

struct Round {                           // buffers and events of one round in flight (two per rank)
    uint64_t* grouped = nullptr;         // world slabs of cap keys, then the spill region (round_reads keys)
    uint32_t* origin = nullptr;          // input position per slot of `grouped`
    uint8_t*  keep_recv = nullptr;       // owner side: flags of what was inserted
    uint8_t*  keep_back = nullptr;       // source side: flags per slot of `grouped`
    uint64_t* grouped_hash = nullptr;    // FQD_SHARD_SEND_HASH: the placement hash of every key slot of `grouped` (slabs only, no spill region)
    uint64_t* recv_hash = nullptr;       //   and, owner side, the hashes of the slabs as they arrived
    uint64_t* d_counts = nullptr;        // CountLayout
    uint64_t* h_counts = nullptr;        // pinned mirror
    uint64_t* d_ins_counts = nullptr;    // owner side, round laid out again: valid keys per sub-slab
    uint64_t* h_ins_counts = nullptr;    // pinned
    uint64_t* slot = nullptr;            // where this round's keys are received (tail of the key store)
    hipEvent_t ev_part = nullptr, ev_xchg = nullptr, ev_spill = nullptr, ev_ins = nullptr, ev_done = nullptr, t0 = nullptr, t1 = nullptr;
    uint64_t n = 0;
    fqd_reads seg[2] = {};               // the round's input (it must stay where it is until the round's flags are final)
    uint8_t* keep_dst = nullptr;
    bool used = false;
    bool relaid = false;                 // owner side: a spill arrived and the round was laid out again
    bool regrouped = false;              // source side: the round was grouped again (Local::re_grouped holds it)
    std::vector<uint64_t> off;           // owner side: first slot of every source's keys once inserted
    uint64_t* h_bad = nullptr;           // pinned: the engine's first-bad-byte word once this round was encoded
};

struct Local {
    fqd_engine* e = nullptr; int device = 0; int rank = 0;
    hipStream_t es = nullptr, cs = nullptr;
    ncclComm_t comm = nullptr;
    uint64_t* records = nullptr;                          // padded groups: the batch's [hash | key] records
    uint64_t* spill = nullptr; size_t spill_cap = 0;      // rare path, owner: the slabs as received + the spills
    uint64_t* re_grouped = nullptr; uint32_t* re_origin = nullptr; uint64_t* re_counts = nullptr;    // rare path, source: the round grouped again
    Round rb[2];
    fqd_shard_stats st{};
};

} // namespace

struct fqd_shard {
    fqd_shard_config cfg{};
    std::vector<Local> lr;
    uint32_t S = 1, K = 0;
    bool padded = false; uint32_t own_len0 = 0, own_len1 = 0;
    bool send_hash = false;              // every key's hash travels with it: the owners do not hash arrived keys again
    Geometry g;
    uint64_t cap = 0;                    // g.cap()
    CountLayout cl{1, 1};
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
void* at_words_rw(uint64_t* p, uint64_t words) { return p ? p + words : nullptr; }

Round* round_of(fqd_shard* s, int rank, uint64_t k) { const int l = local_of(s, rank); return l >= 0 ? &s->lr[size_t(l)].rb[k & 1] : nullptr; }

// Queues round k's all-to-all: slabs out, counts beside them.
int exchange_forward(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const CountLayout& cl = s->cl;
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
                          b ? at_words_rw(b->slot, fqd_plan::slab_slot(uint32_t(src), s->cap) * s->K) : nullptr, slab_words * 8});
            if (s->send_hash)
                xs.push_back({src, dst, a ? at_words(a->grouped_hash, fqd_plan::slab_slot(uint32_t(dst), s->cap)) : nullptr,
                              b ? at_words_rw(b->recv_hash, fqd_plan::slab_slot(uint32_t(src), s->cap)) : nullptr, s->cap * 8});
            xs.push_back({src, dst, a ? at_words(a->d_counts, cl.oc() + uint64_t(dst) * cl.G) : nullptr,
                          b ? at_words_rw(b->d_counts, cl.ic() + uint64_t(src) * cl.G) : nullptr, cl.G * 8});
            xs.push_back({src, dst, a ? at_words(a->d_counts, cl.ot() + uint64_t(dst)) : nullptr,
                          b ? at_words_rw(b->d_counts, cl.it() + uint64_t(src)) : nullptr, 8});
            xs.push_back({src, dst, a ? at_words(a->d_counts, cl.of()) : nullptr,
                          b ? at_words_rw(b->d_counts, cl.il() + uint64_t(src)) : nullptr, 8});
        }
    int rc = move(s, xs, ready);
    if (rc) return rc;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipEventRecord(r.t1, l.cs));
        SH_HIP(s, hipMemcpyAsync(r.h_counts + cl.ic(), r.d_counts + cl.ic(), size_t(cl.words() - cl.ic()) * 8, hipMemcpyDeviceToHost, l.cs));
        SH_HIP(s, hipEventRecord(r.ev_xchg, l.cs));
        l.st.slab_records = s->cap;
    }
    return FQD_OK;
}

// Did a sub-slab of this pair overflow in a one-pass grouping (so that the source sends the slab again)?
bool pair_resends(const uint64_t* chunk_counts, uint64_t layout_classic, const Geometry& g)
{
    if (layout_classic) return false;
    for (uint32_t c = 0; c < g.chunks; ++c) if (chunk_counts[c] > g.sub_cap) return true;
    return false;
}

// The owner side of round k once its keys have arrived: insert (after settling whatever did not fit).
int finish_receive(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const Geometry& g = s->g;
    const CountLayout& cl = s->cl;
    const uint64_t cap = s->cap, K = s->K, G = g.chunks;
    // the host reads the counts here — long after the exchange was queued, with the next encoder already behind it
    bool any = false;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        SH_HIP(s, hipEventSynchronize(r.ev_xchg));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.t0, r.t1) == hipSuccess) l.st.exchange_ms += ms;
        r.relaid = false; r.regrouped = false;
        bool mine = false;
        for (int p = 0; p < W; ++p) {
            if (r.h_counts[cl.it() + p] > cap) r.relaid = true;
            if (pair_resends(r.h_counts + cl.oc() + uint64_t(p) * G, r.h_counts[cl.of()], g)) r.regrouped = true;
            mine = mine || r.h_counts[cl.ot() + p] > cap || r.h_counts[cl.it() + p] > cap ||
                   pair_resends(r.h_counts + cl.ic() + uint64_t(p) * G, r.h_counts[cl.il() + p], g);
        }
        mine = mine || r.regrouped;
        if (mine) l.st.overflow_rounds++;
        any = any || mine;
    }
    if (any) {
        std::vector<Xfer> xs;
        std::vector<hipEvent_t> ready;
        for (Local& l : s->lr) {
            Round& r = l.rb[k & 1];
            SH_HIP(s, hipSetDevice(l.device));
            if (r.regrouped) {
                // ---- source: a sub-slab overflowed in the one-pass grouping, which leaves such keys out.  The round is
                // grouped again the three-step way — slabs filled from their first slot on, spill region written — into
                // buffers of its own: what other owners already hold in the first layout stays as they hold it
                const uint64_t slots = uint64_t(W) * cap + s->cfg.round_reads;
                if (!l.re_grouped) {
                    SH_HIP(s, hipMalloc(reinterpret_cast<void**>(&l.re_grouped), slots * K * 8));
                    SH_HIP(s, hipMalloc(reinterpret_cast<void**>(&l.re_origin), slots * 4));
                    SH_HIP(s, hipMalloc(reinterpret_cast<void**>(&l.re_counts), (uint64_t(W) * G + W + 1) * 8));
                }
                SH_ENG(s, l, fqd_encode_slabs(l.e, r.seg, r.n, uint32_t(W), g.chunk_reads, g.chunks, g.sub_cap, l.re_grouped,
                                              l.re_counts, l.re_counts + uint64_t(W) * G, l.re_origin, FQD_SLABS_EXACT));
                // the flags of the owners that get the slab again come back in ITS slot order: their part of origin[] follows
                for (int p = 0; p < W; ++p)
                    if (pair_resends(r.h_counts + cl.oc() + uint64_t(p) * G, 0, g))
                        SH_HIP(s, hipMemcpyAsync(r.origin + uint64_t(p) * cap, l.re_origin + uint64_t(p) * cap, cap * 4, hipMemcpyDeviceToDevice, l.es));
                const uint64_t spilled = fqd_plan::spill_slot(r.h_counts + cl.ot(), uint32_t(W), uint32_t(W), cap) - uint64_t(W) * cap;
                if (spilled) SH_HIP(s, hipMemcpyAsync(r.origin + uint64_t(W) * cap, l.re_origin + uint64_t(W) * cap, spilled * 4, hipMemcpyDeviceToDevice, l.es));
                SH_HIP(s, hipEventRecord(r.ev_part, l.es));
            }
            ready.push_back(r.ev_part);                        // the sources' send buffers are complete behind this
            if (!r.relaid) continue;
            // ---- owner: a source's slab had no room for everything: [all slabs as received][spill of source 0][of source 1]... in a buffer of its own
            const size_t need = fqd_plan::spill_slot(r.h_counts + cl.it(), uint32_t(W), uint32_t(W), cap) * K * 8;
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
                const int la = local_of(s, src), lb = local_of(s, dst);
                const uint64_t total = a ? a->h_counts[cl.ot() + dst] : b->h_counts[cl.it() + src];
                const bool again = a ? pair_resends(a->h_counts + cl.oc() + uint64_t(dst) * G, a->h_counts[cl.of()], g)
                                     : pair_resends(b->h_counts + cl.ic() + uint64_t(src) * G, b->h_counts[cl.il() + src], g);
                const uint64_t* keys = a ? (a->regrouped ? s->lr[size_t(la)].re_grouped : a->grouped) : nullptr;
                uint64_t* land = b ? (b->relaid ? s->lr[size_t(lb)].spill : b->slot) : nullptr;
                if (again)                                     // the whole slab once more, now filled from its first slot on
                    xs.push_back({src, dst, at_words(keys, fqd_plan::slab_slot(uint32_t(dst), cap) * K), at_words_rw(land, fqd_plan::slab_slot(uint32_t(src), cap) * K), cap * K * 8});
                if (total > cap)                               // and what the slab has no room for
                    xs.push_back({src, dst, a ? at_words(keys, fqd_plan::spill_slot(a->h_counts + cl.ot(), uint32_t(W), uint32_t(dst), cap) * K) : nullptr,
                                  b ? at_words_rw(s->lr[size_t(lb)].spill, fqd_plan::spill_slot(b->h_counts + cl.it(), uint32_t(W), uint32_t(src), cap) * K) : nullptr,
                                  (total - cap) * K * 8});
            }
        if (!xs.empty()) { const int rc = move(s, xs, ready); if (rc) return rc; }
        for (Local& l : s->lr) { SH_HIP(s, hipSetDevice(l.device)); SH_HIP(s, hipEventRecord(l.rb[k & 1].ev_spill, l.cs)); }
    }
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        r.off.assign(size_t(W) + 1, 0);
        // valid keys per sub-slab as the owner will insert them; a source that sent its slab again filled it from the first slot on
        bool fixed = false;
        uint64_t n_sub = 0;
        for (int p = 0; p < W; ++p) {
            const uint64_t total = r.h_counts[cl.it() + p];
            const bool again = pair_resends(r.h_counts + cl.ic() + uint64_t(p) * G, r.h_counts[cl.il() + p], g);
            const uint64_t subs = r.relaid ? fqd_plan::owner_sub_slabs(total, g) : G;
            r.off[size_t(p)] = n_sub * g.sub_cap;
            for (uint64_t c = 0; c < subs; ++c) {
                uint64_t v;
                if (c < G && !again && !r.h_counts[cl.il() + p]) v = r.h_counts[cl.ic() + uint64_t(p) * G + c];       // as the one-pass encoder filled it
                else if (c < G) v = std::min<uint64_t>(fqd_plan::classic_count(total, uint32_t(c), g.chunks, g.sub_cap), g.sub_cap);
                else v = std::min<uint64_t>(total - cap - (c - G) * g.sub_cap, g.sub_cap);                               // a sub-slab of the spill
                r.h_ins_counts[n_sub + c] = v;
            }
            fixed = fixed || again || r.h_counts[cl.il() + p] != 0;      // (a slab filled from its first slot on: its sub-slab counts did not travel)
            n_sub += subs;
        }
        r.off[size_t(W)] = n_sub * g.sub_cap;
        if (!r.relaid) {
            SH_HIP(s, hipStreamWaitEvent(l.es, any ? r.ev_spill : r.ev_xchg, 0));
            const uint64_t* counts = r.d_counts + cl.ic();
            if (fixed) {                                       // a slab came again: its sub-slab counts are not the ones that travelled
                SH_HIP(s, hipMemcpyAsync(r.d_ins_counts, r.h_ins_counts, n_sub * 8, hipMemcpyHostToDevice, l.es));
                counts = r.d_ins_counts;
            }
            // the hashes that came with the keys serve as long as every slab lies the way the one-pass encoder filled it; a slab
            // that came again, or came filled from its first slot on, is hashed here (same function, same values)
            if (s->send_hash && !fixed) SH_ENG(s, l, fqd_insert_slabs_hashed(l.e, r.slot, r.recv_hash, uint32_t(n_sub), g.sub_cap, counts, s->own_len0, s->own_len1, r.keep_recv));
            else                        SH_ENG(s, l, fqd_insert_slabs(l.e, r.slot, uint32_t(n_sub), g.sub_cap, counts, s->own_len0, s->own_len1, r.keep_recv));
        } else {
            // this owner received a spill: every source's slab, then its spill as further sub-slabs, in (source, position) order
            SH_HIP(s, hipStreamWaitEvent(l.es, r.ev_spill, 0));
            uint64_t* slot = nullptr;
            SH_ENG(s, l, fqd_reserve_keys(l.e, n_sub * g.sub_cap, s->own_len0, s->own_len1, &slot));
            uint64_t spill_at = uint64_t(W) * cap;
            for (int p = 0; p < W; ++p) {
                const uint64_t total = r.h_counts[cl.it() + p];
                SH_HIP(s, hipMemcpyAsync(slot + r.off[size_t(p)] * K, l.spill + uint64_t(p) * cap * K, cap * K * 8, hipMemcpyDeviceToDevice, l.es));
                if (total > cap) {
                    SH_HIP(s, hipMemcpyAsync(slot + (r.off[size_t(p)] + cap) * K, l.spill + spill_at * K, (total - cap) * K * 8, hipMemcpyDeviceToDevice, l.es));
                    spill_at += total - cap;
                }
            }
            SH_HIP(s, hipMemcpyAsync(r.d_ins_counts, r.h_ins_counts, n_sub * 8, hipMemcpyHostToDevice, l.es));
            SH_ENG(s, l, fqd_insert_slabs(l.e, slot, uint32_t(n_sub), g.sub_cap, r.d_ins_counts, s->own_len0, s->own_len1, r.keep_recv));
        }
        SH_HIP(s, hipEventRecord(r.ev_ins, l.es));
    }
    return FQD_OK;
}

// Flags of round k back to where the reads came from, and into input order.
int return_flags(fqd_shard* s, uint64_t k)
{
    const int W = s->cfg.world;
    const CountLayout& cl = s->cl;
    const uint64_t cap = s->cap;
    std::vector<Xfer> xs;
    std::vector<hipEvent_t> ready;
    for (Local& l : s->lr) ready.push_back(l.rb[k & 1].ev_ins);
    for (int own = 0; own < W; ++own)
        for (int src = 0; src < W; ++src) {
            Round* o = round_of(s, own, k); Round* a = round_of(s, src, k);
            if (!o && !a) continue;
            // a source's flags start where the owner put its slab (slot src*cap, or further on in a round laid out again —
            // only the owner needs to know); the flags of its spill follow the slab's
            const uint64_t total = a ? a->h_counts[cl.ot() + own] : o->h_counts[cl.it() + src];
            const uint8_t* from = o ? o->keep_recv + o->off[size_t(src)] : nullptr;
            xs.push_back({own, src, from, a ? a->keep_back + fqd_plan::slab_slot(uint32_t(own), cap) : nullptr, cap});   // fixed size; unused slots' flags mean nothing
            if (total > cap)
                xs.push_back({own, src, o ? from + cap : nullptr,
                              a ? a->keep_back + fqd_plan::spill_slot(a->h_counts + cl.ot(), uint32_t(W), uint32_t(own), cap) : nullptr, total - cap});
        }
    int rc = move(s, xs, ready);
    if (rc) return rc;
    for (Local& l : s->lr) {
        Round& r = l.rb[k & 1];
        SH_HIP(s, hipSetDevice(l.device));
        const uint64_t slots = fqd_plan::spill_slot(r.h_counts + cl.ot(), uint32_t(W), uint32_t(W), cap);
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
            if (r.grouped_hash) (void)hipFree(r.grouped_hash);
            if (r.recv_hash) (void)hipFree(r.recv_hash);
            if (r.d_counts) (void)hipFree(r.d_counts);
            if (r.d_ins_counts) (void)hipFree(r.d_ins_counts);
            if (r.h_counts) (void)hipHostFree(r.h_counts);
            if (r.h_ins_counts) (void)hipHostFree(r.h_ins_counts);
            if (r.h_bad) (void)hipHostFree(r.h_bad);
            for (hipEvent_t ev : {r.ev_part, r.ev_xchg, r.ev_spill, r.ev_ins, r.ev_done, r.t0, r.t1}) if (ev) (void)hipEventDestroy(ev);
        }
        if (l.records) (void)hipFree(l.records);
        if (l.spill) (void)hipFree(l.spill);
        if (l.re_grouped) (void)hipFree(l.re_grouped);
        if (l.re_origin) (void)hipFree(l.re_origin);
        if (l.re_counts) (void)hipFree(l.re_counts);
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

uint64_t fqd_shard_slab_capacity(uint64_t round_reads, int32_t world, uint32_t)
{
    if (world <= 0) return 0;
    return fqd_plan::geometry(round_reads, uint32_t(world)).cap();
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
    s->send_hash = (cfg->flags & FQD_SHARD_SEND_HASH) != 0 && !s->padded;
    s->K = s->padded ? fqd_padded_key_words(cfg->len0, cfg->len1) : fqd_key_words(cfg->len0, cfg->len1);
    s->S = cfg->len1 ? 2u : 1u;
    s->own_len0 = s->padded ? s->K : cfg->len0;                   // what the owners' engines are told their keys are
    s->own_len1 = s->padded ? FQD_OPAQUE_KEYS : cfg->len1;
    s->g = fqd_plan::geometry(cfg->round_reads, uint32_t(cfg->world), cfg->slab_records);
    s->cap = s->g.cap();
    s->cl = CountLayout{uint64_t(cfg->world), uint64_t(s->g.chunks)};
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
    // sub-slabs an owner may have to list when it lays a round out again: every slab, and every spill as further sub-slabs
    const uint64_t ins_subs = W * s->g.chunks + (W * cfg->round_reads + s->g.sub_cap - 1) / s->g.sub_cap + W;
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
                (err = hipMalloc(reinterpret_cast<void**>(&r.keep_recv), ins_subs * s->g.sub_cap + s->cap)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.keep_back), slots + s->cap)) != hipSuccess ||
                (s->send_hash && (err = hipMalloc(reinterpret_cast<void**>(&r.grouped_hash), W * s->cap * 8)) != hipSuccess) ||
                (s->send_hash && (err = hipMalloc(reinterpret_cast<void**>(&r.recv_hash), W * s->cap * 8)) != hipSuccess) ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.d_counts), s->cl.words() * 8)) != hipSuccess ||
                (err = hipMalloc(reinterpret_cast<void**>(&r.d_ins_counts), ins_subs * 8)) != hipSuccess ||
                (err = hipHostMalloc(reinterpret_cast<void**>(&r.h_counts), s->cl.words() * 8, hipHostMallocDefault)) != hipSuccess ||
                (err = hipHostMalloc(reinterpret_cast<void**>(&r.h_ins_counts), ins_subs * 8, hipHostMallocDefault)) != hipSuccess ||
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
        r.used = true; r.n = n[a]; r.keep_dst = keep[a]; r.relaid = false; r.regrouped = false;
        for (uint32_t m = 0; m < s->S; ++m) r.seg[m] = seg[a * s->S + m];
        const CountLayout& cl = s->cl;
        if (s->padded) {
            SH_ENG(s, l, fqd_encode_padded(l.e, seg + a * s->S, n[a], s->cfg.len0, s->cfg.len1, l.records));
            SH_ENG(s, l, fqd_partition_slabs(l.e, l.records, n[a], s->K, uint32_t(W), s->cap, r.grouped, r.d_counts + cl.ot(), r.origin));
            SH_HIP(s, hipMemsetAsync(r.d_counts + cl.of(), 0, 8, l.es));
            SH_HIP(s, hipMemsetAsync(r.d_counts + cl.of(), 1, 1, l.es));       // layout: slabs filled from their first slot on
        } else {
            // one pass where it applies (it says which way it went in the word behind the totals); a sub-slab that overflows is
            // noticed from the counts and the round grouped again (finish_receive)
            SH_ENG(s, l, fqd_encode_slabs_hashed(l.e, seg + a * s->S, n[a], uint32_t(W), s->g.chunk_reads, s->g.chunks, s->g.sub_cap, r.grouped,
                                                 s->send_hash ? r.grouped_hash : nullptr, r.d_counts + cl.oc(), r.d_counts + cl.ot(), r.origin, 0u));
        }
        SH_HIP(s, hipMemcpyAsync(r.h_counts, r.d_counts, size_t(cl.ic()) * 8, hipMemcpyDeviceToHost, l.es));
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
