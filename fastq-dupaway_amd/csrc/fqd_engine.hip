// fqd_engine.hip — the C-ABI library (include/fqdupaway.h) over the gfx950 kernels.
//
// One engine = one HBM-resident exact set of sequence keys on one MI355X:
//   table  : open addressing, 8-byte slots (tag:32 | first record index:32), <= 50 % load
//   keys   : packed key words of EVERY record submitted so far (64 B per 150-bp read),
//            so a tag match can always be verified word-for-word
// It stands where the reference keeps `std::unordered_set<setRecord>` plus its
// find/insert loop (hash_dup_remover.hpp:70-71,113-144,206-248).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fqdupaway.h"
#include "fqd_kernels.hpp"

using namespace fqd;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

enum Kind { K_ENCODE = 0, K_INSERT = 1, K_OTHER = 2, K_PARTITION = 3, K_DEDUP = 4 };

struct Timed { hipEvent_t a, b; int kind; uint64_t reads; };

} // namespace

struct fqd_engine {
    int          device = 0;
    int          S = 1;
    uint32_t     flags = 0;
    hipStream_t  stream = nullptr;
    bool         own_stream = false;
    hipStream_t  aux = nullptr;                      // encoder side of the encode/insert overlap
    std::vector<hipEvent_t> sync_events;             // untimed events for cross-stream ordering
    size_t       sync_next = 0;
    hipEvent_t   order_event = nullptr;              // fqd_engine_wait_stream / fqd_stream_wait_engine
    int          n_cu = 256;
    uint64_t     chunk_reads = 8u << 20;             // sub-batch of the overlapped pipeline
    uint32_t     enc_blocks_per_cu = 4, ins_blocks_per_cu = 4;

    DevBuf   table;    uint64_t slots = 0;
    uint32_t seg_bits = 0;                           // log2(slots per probing segment)
    uint32_t tag_mask = 0;                           // slot tag = (hash >> 32) & tag_mask (slot_tag)
    const uint64_t* hashed_records = nullptr;        // fqd_encode_uniform's last output whose hashes still lie in `hashes`
    uint64_t hashed_n = 0; uint32_t hashed_rec_words = 0;
    bool     table_exact = false;                    // sized once for a known total (capacity hint): may run denser than 50 %
    bool     table_clear = false;                    // every slot is EMPTY right now
    bool     table_stale = false;                    // contents are garbage: clear (or rebuild) before use
    bool     finalised = false;                      // fqd_submit_final was called: the table was not written back; only a reset reopens the engine
    DevBuf   bulk_recs, bulk_meta;                   // scratch of the bulk (partitioned) insert
    uint64_t bulk_min = 1u << 20;                    // batches at least this large take the bulk path
    DevBuf   keys;     uint64_t keys_used = 0;       // words
    DevBuf   koff;                                   // ragged only: word offset per record
    bool     ragged = false, have_shape = false;
    uint32_t L0 = 0, L1 = 0, W0 = 0;
    uint64_t n_records = 0;
    uint64_t cap_hint_reads = 0, cap_hint_bases = 0;

    DevBuf   hashes;                                 // per-batch placement hashes
    DevBuf   pad_koff;                               // fqd_encode_padded: slot offsets of a batch
    DevBuf   slab_records;                           // fqd_encode_slabs, three-step path: the batch's [hash | key] records
    DevBuf   scan_scratch;
    DevBuf   part_scratch;
    DevBuf   st_bases[2], st_off[2], st_len[2], st_keep;   // staging for host-space submits
    uint64_t* d_state = nullptr;                     // [0] error word, [1] dups, [2] table-full, [3] scratch, [4] reads longer than a padded key slot
    uint64_t* h_state = nullptr;                     // pinned mirror

    std::string last_error;
    bool     has_bad = false;
    uint64_t bad_record = 0; uint32_t bad_seg = 0, bad_pos = 0; uint8_t bad_byte = 0;

    std::vector<Timed>      pending;
    std::vector<hipEvent_t> free_events;
    fqd_profile prof{};

    int fail(int code, const std::string& msg) { last_error = msg; return code; }
    int fail_hip(const char* what, hipError_t err)
    {
        last_error = std::string(what) + ": " + hipGetErrorString(err);
        (void)hipGetLastError();       // the runtime remembers the failure: a later launch check must not find it again
        return FQD_ERR_HIP;
    }
};

#define HIP_TRY(e, expr)                                                       \
    do { hipError_t _err = (expr); if (_err != hipSuccess) return (e)->fail_hip(#expr, _err); } while (0)

namespace {

inline uint64_t pow2_at_least(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }
inline uint32_t grid_for(const fqd_engine* e, uint64_t n, uint32_t per_block = kBlock)
{
    const uint64_t want = (n + per_block - 1) / per_block;
    const uint64_t cap = uint64_t(e->n_cu) * 8u;
    return uint32_t(std::max<uint64_t>(1, std::min(want, cap)));
}

// Grows a buffer to `bytes`; when `used` > 0 the first `used` bytes are carried over.
int reserve(fqd_engine* e, DevBuf& b, size_t bytes, size_t used = 0)
{
    if (bytes <= b.cap) return FQD_OK;
    size_t want = std::max(bytes, b.cap + b.cap / 2);
    want = (want + 255) & ~size_t(255);
    void* np = nullptr;
    // FQD_ENGINE_TIMING: allocations that took long, to stderr (a hipMalloc waits while the driver clears pages another
    // process has just given back — seconds for tens of gigabytes)
    static const bool timing = std::getenv("FQD_ENGINE_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(e, hipMalloc(&np, want));
    if (timing) {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > 0.005) std::fprintf(stderr, "[engine timing] hipMalloc of %.1f MB took %.3f s\n", double(want) / 1e6, dt);
    }
    if (used && b.p) {
        hipError_t err = hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, e->stream);
        if (err != hipSuccess) { (void)hipFree(np); return e->fail_hip("hipMemcpyAsync(grow)", err); }
    }
    if (b.p) {
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        HIP_TRY(e, hipFree(b.p));
    }
    b.p = np; b.cap = want;
    return FQD_OK;
}

void release(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

// ---- profiling brackets ---------------------------------------------------------
hipEvent_t take_event(fqd_engine* e)
{
    if (!e->free_events.empty()) { hipEvent_t ev = e->free_events.back(); e->free_events.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}
struct Bracket {
    fqd_engine* e; Timed t; bool on; hipStream_t s;
    Bracket(fqd_engine* eng, int kind, uint64_t reads, hipStream_t on_stream = nullptr)
        : e(eng), on((eng->flags & FQD_FLAG_PROFILE) != 0), s(on_stream ? on_stream : eng->stream)
    {
        if (!on) return;
        t.a = take_event(e); t.b = take_event(e); t.kind = kind; t.reads = reads;
        (void)hipEventRecord(t.a, s);
    }
    ~Bracket() { if (on) { (void)hipEventRecord(t.b, s); e->pending.push_back(t); } }
};
hipEvent_t next_sync_event(fqd_engine* e)
{
    if (e->sync_next == e->sync_events.size()) {
        hipEvent_t ev = nullptr;
        (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        e->sync_events.push_back(ev);
    }
    return e->sync_events[e->sync_next++];
}
void drain_profile(fqd_engine* e)
{
    for (const Timed& t : e->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            if (t.kind == K_ENCODE)      { e->prof.encode_ms += ms; e->prof.encode_launches++; e->prof.encode_reads += t.reads; }
            else if (t.kind == K_INSERT) { e->prof.insert_ms += ms; e->prof.insert_launches++; e->prof.insert_reads += t.reads; }
            else if (t.kind == K_PARTITION) { e->prof.partition_ms += ms; e->prof.partition_launches++; e->prof.partition_reads += t.reads; }
            else if (t.kind == K_DEDUP) { e->prof.dedup_ms += ms; e->prof.dedup_launches++; e->prof.dedup_reads += t.reads; }
            else                         { e->prof.other_ms += ms;  e->prof.other_launches++; }
        }
        e->free_events.push_back(t.a); e->free_events.push_back(t.b);
    }
    e->pending.clear();
}

// ---- exclusive scan of uint64 (in place), returns device pointer to the grand total ----
int scan_exclusive(fqd_engine* e, uint64_t* data, uint64_t n, uint64_t add, const uint64_t** total_out)
{
    // level sizes
    std::vector<uint64_t> sizes; sizes.push_back(n);
    while (sizes.back() > 1) sizes.push_back((sizes.back() + kScanTile - 1) / kScanTile);
    if (sizes.size() == 1) sizes.push_back(1);        // n == 1: still one tile
    uint64_t scratch_words = 0;
    for (size_t l = 1; l < sizes.size(); ++l) scratch_words += sizes[l];
    scratch_words += 1;
    int rc = reserve(e, e->scan_scratch, scratch_words * sizeof(uint64_t));
    if (rc) return rc;
    std::vector<uint64_t*> lvl(sizes.size());
    lvl[0] = data;
    uint64_t* s = e->scan_scratch.as<uint64_t>();
    for (size_t l = 1; l < sizes.size(); ++l) { lvl[l] = s; s += sizes[l]; }
    uint64_t* top_total = s;
    // upsweep: scan tiles of each level, totals become the next level
    for (size_t l = 0; l + 1 < sizes.size(); ++l) {
        const uint32_t tiles = uint32_t((sizes[l] + kScanTile - 1) / kScanTile);
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(kBlock), 0, e->stream, lvl[l], sizes[l], lvl[l + 1]);
    }
    // the last level has one element = grand total; keep a copy, then make it an exclusive prefix (0)
    HIP_TRY(e, hipMemcpyAsync(top_total, lvl[sizes.size() - 1], sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(lvl[sizes.size() - 1], 0, sizeof(uint64_t), e->stream));
    // downsweep
    for (size_t l = sizes.size() - 1; l-- > 0;) {
        const uint64_t a = (l == 0) ? add : 0;
        hipLaunchKernelGGL(scan_add_kernel, dim3(grid_for(e, sizes[l])), dim3(kBlock), 0, e->stream,
                           lvl[l], sizes[l], static_cast<const uint64_t*>(lvl[l + 1]), a);
    }
    HIP_TRY(e, hipGetLastError());
    if (total_out) *total_out = top_total;
    return FQD_OK;
}

KeyStore key_store(const fqd_engine* e)
{
    return KeyStore{e->keys.as<uint64_t>(), e->ragged ? e->koff.as<uint64_t>() : nullptr, e->W0, e->W0, 0};
}

// Probing segments: 4096..16384 slots so that (slots / segment) <= 131072 buckets, or the whole
// table when it is smaller than one segment.
uint32_t seg_bits_for(uint64_t slots)
{
    uint32_t t = 0; while ((1ull << t) < slots) ++t;
    if (t <= 12) return t;
    uint32_t want = 13u;                                 // 64 KiB of LDS per segment: best of the 12/13/14 sweep
    if (const char* v = std::getenv("FQD_SEG_BITS")) want = uint32_t(std::min(14, std::max(12, std::atoi(v))));
    return std::min<uint32_t>(14u, std::max<uint32_t>(want, t >= 17 ? t - 17 : 12u));
}

// How a table of 2^t slots with 2^seg_bits-slot segments is split into partition digits, and
// how wide its slot tags can be so that a partition record fits 8 bytes (fqd_kernels.hpp,
// BulkGeom): seg_bits + bits2 + tag bits = 32.
void table_digits(uint32_t t, uint32_t seg_bits, uint32_t& bits1, uint32_t& bits2)
{
    const uint32_t nb_bits = t > seg_bits ? t - seg_bits : 0;
    bits1 = nb_bits <= 8 ? nb_bits : std::min<uint32_t>(8u, (nb_bits + 1) / 2);   // level 1: 256 ways at most
    bits2 = nb_bits - bits1;                                                      // level 2: 512 ways at most (bulk_plan checks)
}
uint32_t tag_mask_for(uint64_t slots, uint32_t seg_bits)
{
    uint32_t t = 0; while ((1ull << t) < slots) ++t;
    uint32_t bits1, bits2;
    table_digits(t, seg_bits, bits1, bits2);
    const uint32_t tag_bits = 32u - std::min(seg_bits, 14u) - std::min(bits2, 9u);
    return tag_bits >= 32 ? 0xFFFFFFFFu : (1u << tag_bits) - 1u;
}

// Keeps the table at <= 50 % load.  `exact`: size for a known total (capacity hint);
// otherwise grow geometrically so rehashes stay rare.
int ensure_table(fqd_engine* e, uint64_t records_after, bool exact = false)
{
    // Load limit: 50 % (a table grows 4x so rehashes stay rare).  A table sized for a KNOWN total
    // (capacity hint) may run denser, FQD_TABLE_PCT slots per 100 records: segments are probed in
    // LDS by the bulk path, where longer probe runs cost little and every slot not allocated is 8
    // bytes of table the dedup kernel does not write back.
    static const uint64_t exact_pct = [] { const char* v = std::getenv("FQD_TABLE_PCT"); const long x = v ? std::atol(v) : 200; return uint64_t(x < 115 ? 115 : (x > 400 ? 400 : x)); }();
    const uint64_t min_slots = e->table_exact ? (records_after * exact_pct + 99) / 100 : 2 * records_after;
    if (e->slots >= min_slots && e->slots) return FQD_OK;
    e->table_exact = exact;
    const uint64_t want = std::max<uint64_t>(pow2_at_least(exact ? (records_after * exact_pct + 99) / 100 : 4 * records_after), 1ull << 16);
    void* nt = nullptr;
    HIP_TRY(e, hipMalloc(&nt, want * sizeof(uint64_t)));
    const uint32_t new_seg_bits = seg_bits_for(want);
    const uint32_t new_tag_mask = tag_mask_for(want, new_seg_bits);
    if (e->slots && e->n_records) {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(nt, 0xFF, want * sizeof(uint64_t), e->stream));
        const KeyStore ks = key_store(e);
        hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(e, e->slots)), dim3(kBlock), 0, e->stream,
                           e->table.as<uint64_t>(), e->slots, static_cast<uint64_t*>(nt), want - 1,
                           (1ull << new_seg_bits) - 1, ks, e->L0, e->L1, uint32_t(e->S == 2),
                           !(e->flags & FQD_FLAG_WEAK_HASH) ? ~0ull : 0x00000000FFFFFFC0ull,
                           new_tag_mask, reinterpret_cast<unsigned long long*>(e->d_state + 1));
        e->table_clear = false; e->table_stale = false;
    } else {
        e->table_clear = false; e->table_stale = true;       // cleared lazily by whoever uses it first
    }
    if (e->table.p) {
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        HIP_TRY(e, hipFree(e->table.p));
    }
    e->table.p = nt; e->table.cap = want * sizeof(uint64_t); e->slots = want; e->seg_bits = new_seg_bits; e->tag_mask = new_tag_mask;
    return FQD_OK;
}

// Switches a uniform engine to the ragged layout (headers + per-record offsets).
int convert_to_ragged(fqd_engine* e, uint64_t records_after)
{
    const uint64_t cap = std::max<uint64_t>(records_after, e->cap_hint_reads);
    if (e->n_records == 0 || !e->have_shape) {
        int rc = reserve(e, e->koff, cap * sizeof(uint64_t));
        if (rc) return rc;
        e->ragged = true; e->have_shape = true; e->keys_used = 0;
        return FQD_OK;
    }
    DevBuf nk, no;
    const uint64_t words = e->n_records * uint64_t(e->W0 + 1);
    int rc = reserve(e, nk, std::max<uint64_t>(words * 2, 1024) * sizeof(uint64_t));
    if (rc) return rc;
    rc = reserve(e, no, cap * sizeof(uint64_t));
    if (rc) { release(nk); return rc; }
    {
        Bracket br(e, K_OTHER, 0);
        const uint64_t header = uint64_t(e->L0) | (uint64_t(e->L1) << 32);
        hipLaunchKernelGGL(relayout_ragged_kernel, dim3(grid_for(e, words)), dim3(kBlock), 0, e->stream,
                           e->keys.as<uint64_t>(), nk.as<uint64_t>(), no.as<uint64_t>(), e->n_records, e->W0, header);
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    release(e->keys); release(e->koff);
    e->keys = nk; e->koff = no; e->keys_used = words; e->ragged = true;
    return FQD_OK;
}

struct StagedChoice { bool staged; uint32_t R; uint32_t tile0, tile1; uint32_t lds_out; };

StagedChoice choose_staged(const fqd_engine* e, const fqd_reads* seg, bool uniform, const KeyStore& ks)
{
    StagedChoice c{false, 0, 0, 0, 0};
    if (!uniform || (e->flags & FQD_FLAG_NO_STAGE)) return c;
    static const uint32_t r_max = [] { const char* v = std::getenv("FQD_STAGE_R"); const int x = v ? std::atoi(v) : 256; return uint32_t(x >= 64 && x <= 256 ? x / 64 * 64 : 256); }();
    for (uint32_t R = r_max; R >= 64; R -= 64) {
        const uint32_t per_tile = e->S == 2 ? R / 2 : R;     // paired: one lane per mate, R/2 pairs per tile
        uint64_t t0 = (uint64_t(per_tile) * seg[0].uniform_stride + 32 + 15) & ~15ull;
        uint64_t t1 = (e->S == 2) ? ((uint64_t(per_tile) * seg[1].uniform_stride + 32 + 15) & ~15ull) : 0;
        if (t0 + t1 <= 64 * 1024) {
            // keys parked in LDS over the lane's own consumed bytes and streamed out as whole lines
            const uint32_t row_words = ks.W0 + ks.lead;
            const uint32_t row0 = ks.lead + seg_words(seg[0].uniform_len);
            const uint32_t row1 = e->S == 2 ? seg_words(seg[1].uniform_len) : 0;
            bool fits = !ks.koff && ks.stride == row_words && row_words > 1 && seg[0].uniform_stride >= 8u * row0 + 15u;
            if (e->S == 2) fits = fits && seg[1].uniform_stride >= 8u * row1 + 15u;
            c = {true, R, uint32_t(t0), uint32_t(t1), fits ? 1u : 0u};
            return c;
        }
    }
    return c;
}

int launch_encode(fqd_engine* e, const SegView* sv, bool uniform, const fqd_reads* seg, uint64_t n,
                  uint64_t first_idx, const KeyStore& ks, uint64_t* hash_out,
                  hipStream_t stream = nullptr, uint32_t blocks_per_cu = 8, const Hist1* fold = nullptr)
{
    if (!stream) stream = e->stream;
    const uint64_t hash_and = (e->flags & FQD_FLAG_WEAK_HASH) ? 0x00000000FFFFFFC0ull : ~0ull;
    Hist1 h1{nullptr, BulkGeom{0, 0, 0, 0, 0}, hash_and};
    if (fold) { h1 = *fold; h1.hash_and = hash_and; }
    Bracket br(e, K_ENCODE, n, stream);
    const StagedChoice c = choose_staged(e, seg, uniform, ks);
    uint64_t* err = e->d_state;
    if (c.staged) {
        const uint32_t per_tile = e->S == 2 ? c.R / 2 : c.R;
        const uint32_t grid = uint32_t(std::min<uint64_t>((n + per_tile - 1) / per_tile, uint64_t(e->n_cu) * blocks_per_cu));
        // FQD_ENC_EXTRA_LDS: bytes of LDS a workgroup asks for on top of its tile (A/B only: what the 12-16 KB of record queues
        // of an encoder that partitions as it goes would cost in resident workgroups — profiles/r03_ab_encoder_lds.jsonl)
        static const size_t extra_lds = [] { const char* v = std::getenv("FQD_ENC_EXTRA_LDS"); return v ? size_t(std::max(0, std::atoi(v))) : size_t(0); }();
        const size_t lds = size_t(c.tile0) + c.tile1 + extra_lds;
        const uint32_t rw = ks.W0 + ks.lead;
        const uint32_t magic = rw > 1 ? uint32_t(((1ull << 32) + rw - 1) / rw) : 0xFFFFFFFFu;   // x/rw for x < 2^16
        if (e->S == 1) {
            // rows of 2^k >= 2 words whose slots are 16-byte aligned: magic 0 selects the kernel's 16-bytes-per-lane stream-out
            static const bool wide_out = [] { const char* v = std::getenv("FQD_ENC_STREAMOUT"); return !(v && v[0] == '0'); }();   // 0: round 3's 8 bytes per lane (A/B)
            const bool pow2_rows = wide_out && rw >= 2 && (rw & (rw - 1)) == 0 && (reinterpret_cast<uintptr_t>(ks.keys) & 15u) == 0;
            const uint32_t magic1 = (c.lds_out && pow2_rows) ? 0u : magic;
            auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(c.R), lds, stream, sv[0], n, first_idx, ks, hash_out, err, magic1, h1);
            };
            if (c.lds_out) launch(encode_staged_kernel<true>); else launch(encode_staged_kernel<false>);
        } else {
            const uint32_t split = ks.lead + seg_words(seg[0].uniform_len);
            static const bool wide_out = [] { const char* v = std::getenv("FQD_ENC_STREAMOUT"); return !(v && v[0] == '0'); }();
            const bool pow2_rows = wide_out && rw >= 2 && (rw & (rw - 1)) == 0 && (split & 1u) == 0 && (reinterpret_cast<uintptr_t>(ks.keys) & 15u) == 0;
            const uint32_t magic2 = (c.lds_out && pow2_rows) ? 0u : magic;
            auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(c.R), lds, stream,
                                   sv[0], sv[1], n, first_idx, ks, hash_out, err, c.tile0, magic2, h1);
            };
            if (c.lds_out) launch(encode_staged_pe_kernel<true>); else launch(encode_staged_pe_kernel<false>);
        }
    } else if (!(e->flags & FQD_FLAG_NO_STAGE) && sv[0].offsets && (e->S == 1 || sv[1].offsets)) {
        // ragged descriptors: stage each tile's span of the input through LDS where the records lie close together
        const uint32_t R = 128, span_cap = e->S == 2 ? 24u * 1024u : 48u * 1024u;
        const uint32_t grid = uint32_t(std::min<uint64_t>((n + R - 1) / R, uint64_t(e->n_cu) * blocks_per_cu));
        if (e->S == 1)
            hipLaunchKernelGGL(encode_span_kernel<1>, dim3(grid), dim3(R), span_cap, stream,
                               sv[0], sv[1], n, first_idx, ks, hash_out, err, span_cap, h1);
        else
            hipLaunchKernelGGL(encode_span_kernel<2>, dim3(grid), dim3(R), 2 * span_cap, stream,
                               sv[0], sv[1], n, first_idx, ks, hash_out, err, span_cap, h1);
    } else {
        const uint32_t grid = uint32_t(std::min<uint64_t>(grid_for(e, n), uint64_t(e->n_cu) * blocks_per_cu));
        if (e->S == 1)
            hipLaunchKernelGGL(encode_general_kernel<1>, dim3(grid), dim3(kBlock), 0, stream,
                               sv[0], sv[1], n, first_idx, ks, hash_out, err, h1);
        else
            hipLaunchKernelGGL(encode_general_kernel<2>, dim3(grid), dim3(kBlock), 0, stream,
                               sv[0], sv[1], n, first_idx, ks, hash_out, err, h1);
    }
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

int launch_insert(fqd_engine* e, const KeyStore& ks, const uint64_t* hashes, uint32_t hash_stride,
                  uint64_t n, uint64_t first_idx, uint8_t* keep, bool preset_keep = true, uint32_t blocks_per_cu = 8,
                  uint32_t* first = nullptr)
{
    if (e->table_stale) {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(e->table.p, 0xFF, e->slots * sizeof(uint64_t), e->stream));
        e->table_stale = false;
    }
    e->table_clear = false;
    if (preset_keep) {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(keep, 1, n, e->stream));
    }
    Bracket br(e, K_INSERT, n);
    const uint32_t grid = uint32_t(std::min<uint64_t>(grid_for(e, n), uint64_t(e->n_cu) * blocks_per_cu));
    hipLaunchKernelGGL(insert_kernel, dim3(grid), dim3(kBlock), 0, e->stream,
                       e->table.as<uint64_t>(), e->slots - 1, (1ull << e->seg_bits) - 1, ks, hashes, hash_stride, n,
                       Verdicts{keep, first, uint32_t(first_idx)}, e->tag_mask,
                       reinterpret_cast<unsigned long long*>(e->d_state + 1));
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

// Bulk insert (bulk_* kernels): partition the batch by table segment, then one workgroup per
// segment dedups in LDS.  Worth it when the batch is large against the table: every segment
// is rewritten once (and, unless the table is known empty, read once).
bool bulk_applies(const fqd_engine* e, uint64_t n)
{
    if (n < e->bulk_min || e->slots < (1ull << 13)) return false;
    const bool empty = e->n_records == 0;
    // Against a table that already holds records the bulk path pays one read and one write of the whole
    // table (0.76 ms for 2 GiB) on top of 2.9 ms per 100 M records; the atomic path costs 6.3 ms per
    // 100 M.  Measured break-even (tools/split_probe.py): about slots / 14 records.
    static const uint64_t ratio = [] { const char* v = std::getenv("FQD_BULK_RATIO"); const long x = v ? std::atol(v) : 12; return uint64_t(x > 0 ? x : 12); }();
    return empty || n * ratio >= e->slots;
}

struct BulkPlan {
    bool ok = false;
    BulkGeom g{0, 0, 0, 0, 0};
    uint32_t nd1 = 0, n_buckets = 0;
    uint32_t *hist1 = nullptr, *start1 = nullptr, *cursor1 = nullptr, *tile_start1 = nullptr;
    uint32_t *hist2 = nullptr, *start2 = nullptr, *cursor2 = nullptr;
    uint64_t *recA = nullptr, *recB = nullptr;
    uint8_t *digit2 = nullptr;
    uint32_t *heavy_count = nullptr, *heavy_list = nullptr;
};

// Geometry + scratch of one bulk insert; zeroes the counters on the engine's stream.
int bulk_plan(fqd_engine* e, uint64_t n, BulkPlan& p)
{
    uint32_t t = 0; while ((1ull << t) < e->slots) ++t;
    const uint32_t nb_bits = t - e->seg_bits;                // <= 17 by construction of seg_bits (t <= 31)
    p.ok = false;
    if (nb_bits > 17 || nb_bits == 0) return FQD_OK;
    p.g.slot_mask = e->slots - 1; p.g.seg_bits = e->seg_bits; p.g.tag_mask = e->tag_mask;
    table_digits(t, e->seg_bits, p.g.bits1, p.g.bits2);
    p.nd1 = 1u << p.g.bits1; p.n_buckets = 1u << nb_bits;
    int rc;
    const size_t rec_bytes = ((n * sizeof(uint64_t)) + 255) & ~size_t(255);
    const size_t d2_bytes = p.g.bits2 ? ((n * sizeof(uint8_t)) + 255) & ~size_t(255) : 0;
    if ((rc = reserve(e, e->bulk_recs, rec_bytes * (p.g.bits2 ? 2 : 1) + d2_bytes))) return rc;
    // meta: hist1[256] start1[257] cursor1[256] tile_start1[257] heavy_count | hist2[nb] start2[nb+1] cursor2[nb] heavy_list[nb]
    const size_t meta_words = 1100 + 4 * size_t(p.n_buckets) + 16;
    if ((rc = reserve(e, e->bulk_meta, meta_words * sizeof(uint32_t)))) return rc;
    uint32_t* m = e->bulk_meta.as<uint32_t>();
    p.hist1 = m; p.start1 = m + 256; p.cursor1 = m + 520; p.tile_start1 = m + 780;
    p.hist2 = m + 1100; p.start2 = p.hist2 + p.n_buckets; p.cursor2 = p.start2 + p.n_buckets + 4;
    p.heavy_count = m + 1090;                                  // zeroed with the rest of meta
    p.heavy_list = p.cursor2 + p.n_buckets + 4;
    p.recA = e->bulk_recs.as<uint64_t>();
    p.recB = reinterpret_cast<uint64_t*>(e->bulk_recs.as<char>() + rec_bytes);
    p.digit2 = p.g.bits2 ? reinterpret_cast<uint8_t*>(e->bulk_recs.as<char>() + 2 * rec_bytes) : nullptr;
    {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(m, 0, meta_words * sizeof(uint32_t), e->stream));
    }
    p.ok = true;
    return FQD_OK;
}

// hist1_done: the encoder already folded the level-1 histogram into its own pass.
int launch_bulk_insert(fqd_engine* e, const KeyStore& ks, const uint64_t* hashes, uint32_t hash_stride,
                       uint64_t n, uint64_t first_idx, uint8_t* keep, const BulkPlan& p, bool hist1_done,
                       uint32_t* first = nullptr, bool last_batch = false)
{
    const Verdicts verdicts{keep, first, uint32_t(first_idx)};
    const BulkGeom g = p.g;
    const uint32_t nd1 = p.nd1, n_buckets = p.n_buckets;
    uint32_t *hist1 = p.hist1, *start1 = p.start1, *cursor1 = p.cursor1, *tile_start1 = p.tile_start1;
    uint32_t *hist2 = p.hist2, *start2 = p.start2, *cursor2 = p.cursor2;
    uint64_t *recA = p.recA, *recB = p.recB;
    const bool fresh = e->n_records == 0 || e->table_clear || e->table_stale;
    {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(keep, 1, n, e->stream));
    }
    const uint64_t* final_recs = recA; const uint32_t* bstart = start1;
    {
    Bracket part_br(e, K_PARTITION, n);
    uint32_t part_per_cu = 2u;                                  // 75 KB of LDS per 8192-record tile: two fit a CU (1/2/3 swept)
    if (const char* v = std::getenv("FQD_PART_BLOCKS_PER_CU")) part_per_cu = uint32_t(std::min(16, std::max(1, std::atoi(v))));
    const uint32_t part_grid = uint32_t(std::min<uint64_t>((n + kPartTile - 1) / kPartTile, uint64_t(e->n_cu) * part_per_cu));
    if (!hist1_done)
        hipLaunchKernelGGL(bulk_hist1_kernel, dim3(part_grid), dim3(kPartThreads), 0, e->stream, hashes, hash_stride, n, g, hist1);
    hipLaunchKernelGGL(bulk_scan256_kernel, dim3(1), dim3(320), 0, e->stream,
                       static_cast<const uint32_t*>(hist1), nd1, start1, cursor1, tile_start1);
    // (round 3 measured, and dropped, software-pipelined and 512-thread forms of these passes and an encoder that keeps its next
    //  tile's loads in flight: profiles/r03_ab_pipelining.jsonl — no gain; the phase stamps of `make STAMPS=1` show why)
    hipLaunchKernelGGL(bulk_scatter_kernel<1>, dim3(part_grid), dim3(kPartThreads), 0, e->stream,
                       hashes, hash_stride, uint32_t(first_idx), static_cast<const uint64_t*>(nullptr), n, g,
                       static_cast<const uint32_t*>(start1), static_cast<const uint32_t*>(tile_start1), cursor1, recA, g.bits2 > 8 ? static_cast<uint8_t*>(nullptr) : p.digit2);
    if (g.bits2) {
        const uint32_t grid2 = uint32_t(std::min<uint64_t>((n + kPartTile - 1) / kPartTile + nd1, uint64_t(e->n_cu) * part_per_cu));
        const uint32_t hgrid = uint32_t(std::min<uint64_t>((n + kPartTile - 1) / kPartTile + nd1, uint64_t(e->n_cu) * 2u));   // 2 KB of LDS: two 1024-thread blocks fill a CU's wave slots
        if (g.bits2 > 8)
            hipLaunchKernelGGL(bulk_hist2_kernel<true>, dim3(hgrid), dim3(kPartThreads), 0, e->stream,
                               static_cast<const uint8_t*>(nullptr), static_cast<const uint64_t*>(recA), g, static_cast<const uint32_t*>(start1),
                               static_cast<const uint32_t*>(tile_start1), hist2);
        else
            hipLaunchKernelGGL(bulk_hist2_kernel<false>, dim3(hgrid), dim3(kPartThreads), 0, e->stream,
                               static_cast<const uint8_t*>(p.digit2), static_cast<const uint64_t*>(nullptr), g, static_cast<const uint32_t*>(start1),
                               static_cast<const uint32_t*>(tile_start1), hist2);
        hipLaunchKernelGGL(bulk_scan_buckets_kernel, dim3(nd1), dim3(512), 0, e->stream,
                           static_cast<const uint32_t*>(hist2), g.bits2, static_cast<const uint32_t*>(start1), nd1, start2, cursor2);
        hipLaunchKernelGGL(bulk_scatter_kernel<2>, dim3(grid2), dim3(kPartThreads), 0, e->stream,
                           static_cast<const uint64_t*>(nullptr), 0u, 0u, static_cast<const uint64_t*>(recA), n, g,
                           static_cast<const uint32_t*>(start1), static_cast<const uint32_t*>(tile_start1), cursor2, recB,
                           static_cast<uint8_t*>(nullptr));
        final_recs = recB; bstart = start2;
    }
    }
    Bracket br(e, K_DEDUP, n);
    const size_t lds = ((size_t(1) << e->seg_bits) + kDedupChunk + 2) * sizeof(uint64_t);
    uint32_t dthreads = 512;
    if (const char* v = std::getenv("FQD_DEDUP_THREADS")) dthreads = uint32_t(std::min(1024, std::max(64, std::atoi(v))));
    const uint32_t dgrid = std::min<uint32_t>(n_buckets, uint32_t(e->n_cu) * uint32_t(std::max<size_t>(1, (160 * 1024) / lds)));
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(e->d_state + 1);
    uint32_t heavy_above = 4u << e->seg_bits;                   // far beyond what a segment can hold at <= 50 % load
    if (const char* v = std::getenv("FQD_HEAVY_ABOVE")) heavy_above = uint32_t(std::max(0, std::atoi(v)));
    // The last batch of a run (fqd_submit_final): nobody will read the table again, so the segments stay in LDS and the
    // 8 bytes per slot of write-back (2 GiB for the 100 M-read table) are not moved.  FQD_FINAL_WRITE_BACK=1: the A/B switch.
    static const bool force_write_back = [] { const char* v = std::getenv("FQD_FINAL_WRITE_BACK"); return v && v[0] == '1'; }();
    const uint32_t write_back = (last_batch && !force_write_back) ? 0u : 1u;
    auto launch_dedup = [&](auto kernel) -> int {
        if (lds > 64 * 1024) HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(kernel, dim3(dgrid), dim3(dthreads), lds, e->stream,
                           final_recs, bstart, n_buckets, e->table.as<uint64_t>(), e->seg_bits, g.seg_bits + g.bits2, ks, verdicts, counters,
                           heavy_above, p.heavy_count, p.heavy_list, write_back);
        return FQD_OK;
    };
    int drc;
    // verify width: 16 bytes per lane when the uniform keys are whole 16-byte units (FQD_DEDUP_VL=0: the 8-byte form)
    int vl = 0;
    if (!ks.koff && ks.lead == 0 && ks.stride == ks.W0 && (ks.W0 & 1u) == 0) vl = ks.W0 <= 8 ? 4 : (ks.W0 <= 16 ? 8 : 0);
    if (const char* v = std::getenv("FQD_DEDUP_VL")) { if (std::atoi(v) == 0) vl = 0; }
    if (ks.koff)      drc = fresh ? launch_dedup(bucket_dedup_kernel<true, true, 0>) : launch_dedup(bucket_dedup_kernel<false, true, 0>);
    else if (vl == 4) drc = fresh ? launch_dedup(bucket_dedup_kernel<true, false, 4>) : launch_dedup(bucket_dedup_kernel<false, false, 4>);
    else if (vl == 8) drc = fresh ? launch_dedup(bucket_dedup_kernel<true, false, 8>) : launch_dedup(bucket_dedup_kernel<false, false, 8>);
    else              drc = fresh ? launch_dedup(bucket_dedup_kernel<true, false, 0>) : launch_dedup(bucket_dedup_kernel<false, false, 0>);
    if (drc) return drc;
    hipLaunchKernelGGL(heavy_bucket_insert_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream,
                       final_recs, bstart, g, e->table.as<uint64_t>(), ks, verdicts, counters,
                       static_cast<const uint32_t*>(p.heavy_count), static_cast<const uint32_t*>(p.heavy_list));
    HIP_TRY(e, hipGetLastError());
    e->table_clear = false; e->table_stale = !write_back;     // not written back = garbage from here on (the engine is finalised)
    return FQD_OK;
}

// Reads back the state words after the stream is idle and turns them into a status.
int check_state(fqd_engine* e)
{
    HIP_TRY(e, hipMemcpyAsync(e->h_state, e->d_state, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->flags & FQD_FLAG_PROFILE) drain_profile(e);
    if (e->h_state[2] != 0)
        return e->fail(FQD_ERR_HIP, "internal: hash set overflowed its table");
    if (e->h_state[4] != 0)
        return e->fail(FQD_ERR_ARG, "fqd_encode_padded: a read is longer than the key width the caller set (max_len0 / max_len1)");
    if (e->h_state[0] != kNoError) {
        const uint64_t w = e->h_state[0];
        e->has_bad = true;
        e->bad_record = w >> 32; e->bad_seg = uint32_t((w >> 31) & 1u);
        e->bad_pos = uint32_t((w >> 8) & 0x7FFFFFu); e->bad_byte = uint8_t(w & 0xFFu);
        char buf[160];
        std::snprintf(buf, sizeof buf, "unknown character in DNA sequence: byte 0x%02x at record %llu, mate %u, position %u",
                      unsigned(e->bad_byte), static_cast<unsigned long long>(e->bad_record), e->bad_seg + 1, e->bad_pos);
        return e->fail(FQD_ERR_BAD_BASE, buf);
    }
    return FQD_OK;
}

bool seg_is_uniform(const fqd_reads& r) { return r.offsets == nullptr && r.lengths == nullptr; }

} // namespace

// ---- what fqd_join.hip needs from an engine (same library, not exported) -----------------
#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e) { return e->stream; }
FQD_HIDDEN int fqd_internal_device(fqd_engine* e) { return e->device; }
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg) { return e->fail(code, msg); }
FQD_HIDDEN uint64_t* fqd_internal_state(fqd_engine* e) { return e->d_state; }
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out)
{
    DevBuf& b = which == 0 ? e->scan_scratch : e->part_scratch;
    const int rc = reserve(e, b, bytes);
    *out = b.p;
    return rc;
}

// =============================== C ABI ==========================================
extern "C" {

int fqd_abi_version(void) { return FQD_ABI_VERSION; }

int fqd_device_count(int* count)
{
    if (!count) return FQD_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return FQD_ERR_NO_DEVICE; }
    *count = n;
    return FQD_OK;
}

uint32_t fqd_key_words(uint32_t len0, uint32_t len1) { return seg_words(len0) + seg_words(len1); }

const char* fqd_last_error(const fqd_engine* e) { return e ? e->last_error.c_str() : g_create_error.c_str(); }

int fqd_engine_create(const fqd_config* cfg, fqd_engine** out)
{
    if (!cfg || !out || (cfg->segments != 1 && cfg->segments != 2)) { g_create_error = "bad fqd_config"; return FQD_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "no HIP device available for the dedup engine (this library has no CPU path)";
        return FQD_ERR_NO_DEVICE;
    }
    fqd_engine* e = new fqd_engine();
    e->device = cfg->device; e->S = cfg->segments; e->flags = cfg->flags;
    e->cap_hint_reads = cfg->capacity_reads; e->cap_hint_bases = cfg->capacity_bases;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        (void)hipGetLastError();
        delete e; return FQD_ERR_HIP;
    };
    hipError_t err;
    if ((err = hipSetDevice(e->device)) != hipSuccess) return bail("hipSetDevice", err);
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, e->device)) != hipSuccess) return bail("hipGetDeviceProperties", err);
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cfg->stream) { e->stream = static_cast<hipStream_t>(cfg->stream); }
    else {
        if ((err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", err);
        e->own_stream = true;
    }
    if ((err = hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate(aux)", err);
    if (const char* v = std::getenv("FQD_CHUNK_READS")) { const long long x = std::atoll(v); e->chunk_reads = x > 0 ? uint64_t(x) : ~0ull; }
    if (const char* v = std::getenv("FQD_BULK_MIN")) { const long long x = std::atoll(v); e->bulk_min = x >= 0 ? uint64_t(x) : ~0ull; }
    if (const char* v = std::getenv("FQD_ENC_BLOCKS_PER_CU")) e->enc_blocks_per_cu = uint32_t(std::max(1, std::atoi(v)));
    if (const char* v = std::getenv("FQD_INS_BLOCKS_PER_CU")) e->ins_blocks_per_cu = uint32_t(std::max(1, std::atoi(v)));
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->d_state), 8 * sizeof(uint64_t))) != hipSuccess) return bail("hipMalloc(state)", err);
    if ((err = hipHostMalloc(reinterpret_cast<void**>(&e->h_state), 8 * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc(state)", err);
    int rc = fqd_engine_reset(e);
    if (rc == FQD_OK && cfg->capacity_reads) rc = ensure_table(e, cfg->capacity_reads, true);
    if (rc == FQD_OK && cfg->capacity_bases) {
        // 2 bits + 1 mask bit per base, rounded up per read: bases*3/8 bytes plus slack
        const uint64_t words = cfg->capacity_bases * 3 / 64 + 2 * cfg->capacity_reads + 1024;
        rc = reserve(e, e->keys, words * sizeof(uint64_t));
    }
    if (rc != FQD_OK) { g_create_error = e->last_error; fqd_engine_destroy(e); return rc; }
    *out = e;
    return FQD_OK;
}

int fqd_engine_destroy(fqd_engine* e)
{
    if (!e) return FQD_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
#ifdef FQD_STAMPS
    {   // diagnostic build: the phase table of the stamped kernels since the last report, in milliseconds of workgroup time
        unsigned long long t[8][16];
        if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stamps), sizeof t) == hipSuccess) {
            static const char* names[4] = {"scatter1", "scatter2", "dedup", "encode"};
            for (int k = 0; k < 4; ++k) {
                unsigned long long tot = 0; for (int q = 0; q < 16; ++q) tot += t[k][q];
                if (!tot) continue;
                std::fprintf(stderr, "[stamps] {\"kernel\": \"%s\", \"total_wg_ms\": %.3f, \"phase_frac\": [", names[k], double(tot) * 1e-5);
                for (int q = 0; q < 12; ++q) std::fprintf(stderr, "%s%.3f", q ? ", " : "", double(t[k][q]) / double(tot));
                std::fprintf(stderr, "]}\n");
            }
            std::memset(t, 0, sizeof t);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), t, sizeof t);
        }
    }
#endif
    drain_profile(e);
    if (e->aux) (void)hipStreamSynchronize(e->aux);
    for (hipEvent_t ev : e->free_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->sync_events) (void)hipEventDestroy(ev);
    if (e->order_event) (void)hipEventDestroy(e->order_event);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    release(e->table); release(e->keys); release(e->koff); release(e->hashes);
    release(e->scan_scratch); release(e->part_scratch); release(e->pad_koff); release(e->slab_records); release(e->st_keep); release(e->bulk_recs); release(e->bulk_meta);
    for (int s = 0; s < 2; ++s) { release(e->st_bases[s]); release(e->st_off[s]); release(e->st_len[s]); }
    if (e->d_state) (void)hipFree(e->d_state);
    if (e->h_state) (void)hipHostFree(e->h_state);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return FQD_OK;
}

int fqd_engine_reset(fqd_engine* e)
{
    if (!e) return FQD_ERR_ARG;
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->table.p && !e->table_clear) e->table_stale = true;     // cleared (or rebuilt by the bulk path) on first use
    e->h_state[0] = kNoError; e->h_state[1] = 0; e->h_state[2] = 0; e->h_state[3] = 0; e->h_state[4] = 0;
    HIP_TRY(e, hipMemcpyAsync(e->d_state, e->h_state, 5 * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->n_records = 0; e->keys_used = 0; e->ragged = false; e->have_shape = false; e->finalised = false;
    e->L0 = e->L1 = e->W0 = 0; e->has_bad = false; e->last_error.clear();
    return FQD_OK;
}

static const char kFinalised[] = "this engine was given its last batch (fqd_submit_final): fqd_engine_reset before anything else is added";

static int submit_impl(fqd_engine* e, const fqd_reads* seg, uint64_t n, int memory, uint8_t* keep, bool last_batch)
{
    if (!e) return FQD_ERR_ARG;
    if (!seg || (n && !keep) || (memory != FQD_MEM_HOST && memory != FQD_MEM_DEVICE))
        return e->fail(FQD_ERR_ARG, "fqd_submit: bad arguments");
    if (e->finalised) return e->fail(FQD_ERR_ARG, kFinalised);
    if (n == 0) { e->finalised = last_batch; return FQD_OK; }
    if (e->n_records + n > 0xFFFFFFFEull) return e->fail(FQD_ERR_CAPACITY, "more than 2^32-2 records in one engine");
    if (e->have_shape && e->L1 == FQD_OPAQUE_KEYS)
        return e->fail(FQD_ERR_ARG, "fqd_submit: this engine holds opaque keys (fqd_insert_keys / fqd_insert_slabs with FQD_OPAQUE_KEYS)");
    HIP_TRY(e, hipSetDevice(e->device));
    int rc;

    bool uniform = true;
    for (int s = 0; s < e->S; ++s) {
        if (!seg[s].bases && !(seg_is_uniform(seg[s]) && seg[s].uniform_len == 0))
            return e->fail(FQD_ERR_ARG, "fqd_submit: null bases");
        if (!seg_is_uniform(seg[s])) {
            if (!seg[s].offsets || !seg[s].lengths) return e->fail(FQD_ERR_ARG, "fqd_submit: offsets and lengths go together");
            uniform = false;
        }
    }

    // ---- host-space input: stage into device buffers ------------------------------
    SegView sv[2] = {{nullptr, nullptr, nullptr, 0, 0}, {nullptr, nullptr, nullptr, 0, 0}};
    uint64_t host_ragged_words = 0;                      // exact key words when lengths are on the host
    uint8_t* d_keep = keep;
    for (int s = 0; s < e->S; ++s) {
        const fqd_reads& r = seg[s];
        sv[s].ulen = r.uniform_len; sv[s].ustride = r.uniform_stride;
        if (memory == FQD_MEM_DEVICE) { sv[s].bases = r.bases; sv[s].offsets = r.offsets; sv[s].lengths = r.lengths; continue; }
        uint64_t extent = 0;
        if (seg_is_uniform(r)) extent = (n - 1) * uint64_t(r.uniform_stride) + r.uniform_len;
        else for (uint64_t i = 0; i < n; ++i) extent = std::max<uint64_t>(extent, r.offsets[i] + r.lengths[i]);
        if ((rc = reserve(e, e->st_bases[s], extent + 16))) return rc;
        if (extent) HIP_TRY(e, hipMemcpyAsync(e->st_bases[s].p, r.bases, extent, hipMemcpyHostToDevice, e->stream));
        sv[s].bases = e->st_bases[s].as<uint8_t>();
        if (!seg_is_uniform(r)) {
            if ((rc = reserve(e, e->st_off[s], n * sizeof(uint64_t)))) return rc;
            if ((rc = reserve(e, e->st_len[s], n * sizeof(uint32_t)))) return rc;
            HIP_TRY(e, hipMemcpyAsync(e->st_off[s].p, r.offsets, n * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(e, hipMemcpyAsync(e->st_len[s].p, r.lengths, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
            sv[s].offsets = e->st_off[s].as<uint64_t>(); sv[s].lengths = e->st_len[s].as<uint32_t>();
        }
    }
    if (memory == FQD_MEM_HOST) {
        if ((rc = reserve(e, e->st_keep, n))) return rc;
        d_keep = e->st_keep.as<uint8_t>();
        if (!uniform)
            for (uint64_t i = 0; i < n; ++i) {
                host_ragged_words += 1;
                for (int s = 0; s < e->S; ++s)
                    host_ragged_words += seg_words(seg[s].lengths ? seg[s].lengths[i] : seg[s].uniform_len);
            }
    }

    // ---- shape: stay uniform while every record has the same mate lengths ----------
    const uint32_t bl0 = seg[0].uniform_len, bl1 = (e->S == 2) ? seg[1].uniform_len : 0u;
    if (!e->have_shape && uniform) {
        e->have_shape = true; e->ragged = false; e->L0 = bl0; e->L1 = bl1; e->W0 = seg_words(bl0) + seg_words(bl1);
    } else if (!e->ragged && (!uniform || bl0 != e->L0 || bl1 != e->L1)) {
        if ((rc = convert_to_ragged(e, e->n_records + n))) return rc;
    }

    const uint64_t first = e->n_records;
    if ((rc = ensure_table(e, first + n))) return rc;
    e->hashed_records = nullptr;                              // the scratch is about to be reused
    if ((rc = reserve(e, e->hashes, n * sizeof(uint64_t)))) return rc;

    uint64_t new_words = 0;
    if (!e->ragged) {
        new_words = n * uint64_t(e->W0);
        if ((rc = reserve(e, e->keys, std::max<uint64_t>((e->keys_used + new_words), 64) * sizeof(uint64_t),
                          e->keys_used * sizeof(uint64_t)))) return rc;
    } else {
        if ((rc = reserve(e, e->koff, (first + n) * sizeof(uint64_t), first * sizeof(uint64_t)))) return rc;
        uint64_t* need = e->koff.as<uint64_t>() + first;
        {
            Bracket br(e, K_OTHER, 0);
            if (e->S == 1) hipLaunchKernelGGL(slot_words_kernel<1>, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream, sv[0], sv[1], n, need);
            else           hipLaunchKernelGGL(slot_words_kernel<2>, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream, sv[0], sv[1], n, need);
            const uint64_t* total_dev = nullptr;
            if ((rc = scan_exclusive(e, need, n, e->keys_used, &total_dev))) return rc;
            if (uniform)                         new_words = n * uint64_t(1 + seg_words(bl0) + seg_words(bl1));
            else if (memory == FQD_MEM_HOST)     new_words = host_ragged_words;
            else {
                HIP_TRY(e, hipMemcpyAsync(&e->h_state[3], total_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(e, hipStreamSynchronize(e->stream));
                new_words = e->h_state[3];
            }
        }
        if ((rc = reserve(e, e->keys, std::max<uint64_t>((e->keys_used + new_words), 64) * sizeof(uint64_t),
                          e->keys_used * sizeof(uint64_t)))) return rc;
    }

    KeyStore ks{e->keys.as<uint64_t>(), e->ragged ? e->koff.as<uint64_t>() : nullptr, e->W0, e->W0, 0};
    BulkPlan plan;
    if (bulk_applies(e, n) && (rc = bulk_plan(e, n, plan))) return rc;
    if (plan.ok) {
        const Hist1 fold{plan.hist1, plan.g, ~0ull};
        static const bool fold_on = [] { const char* v = std::getenv("FQD_FOLD_HIST1"); return !(v && v[0] == '0'); }();
        if ((rc = launch_encode(e, sv, uniform, seg, n, first, ks, e->hashes.as<uint64_t>(), nullptr, 8, fold_on ? &fold : nullptr))) return rc;
        if ((rc = launch_bulk_insert(e, ks, e->hashes.as<uint64_t>(), 1, n, first, d_keep, plan, fold_on, nullptr, last_batch))) return rc;
    } else if (n >= 2 * e->chunk_reads && e->aux) {
        // Overlap: the encoder streams HBM, the insert is bound by memory-side atomics, so the
        // two run side by side on two streams, sub-batch k+1 being encoded while k is inserted.
        // Each kernel is launched on half of the wave slots so the other one can be resident.
        {
            Bracket br(e, K_OTHER, 0);
            HIP_TRY(e, hipMemsetAsync(d_keep, 1, n, e->stream));
        }
        e->sync_next = 0;
        hipEvent_t start = next_sync_event(e);
        HIP_TRY(e, hipEventRecord(start, e->stream));
        HIP_TRY(e, hipStreamWaitEvent(e->aux, start, 0));
        for (uint64_t a = 0; a < n; a += e->chunk_reads) {
            const uint64_t c = std::min<uint64_t>(e->chunk_reads, n - a);
            SegView sub[2] = {sv[0], sv[1]};
            fqd_reads subseg[2] = {seg[0], seg[1]};
            for (int s2 = 0; s2 < e->S; ++s2) {
                if (sub[s2].offsets) { sub[s2].offsets += a; sub[s2].lengths += a; }
                else sub[s2].bases += a * uint64_t(sub[s2].ustride);
            }
            if ((rc = launch_encode(e, sub, uniform, subseg, c, first + a, ks, e->hashes.as<uint64_t>() + a,
                                    e->aux, e->enc_blocks_per_cu))) return rc;
            hipEvent_t done = next_sync_event(e);
            HIP_TRY(e, hipEventRecord(done, e->aux));
            HIP_TRY(e, hipStreamWaitEvent(e->stream, done, 0));
            if ((rc = launch_insert(e, ks, e->hashes.as<uint64_t>() + a, 1, c, first + a, d_keep + a, false,
                                    e->ins_blocks_per_cu))) return rc;
        }
    } else {
        if ((rc = launch_encode(e, sv, uniform, seg, n, first, ks, e->hashes.as<uint64_t>()))) return rc;
        if ((rc = launch_insert(e, ks, e->hashes.as<uint64_t>(), 1, n, first, d_keep))) return rc;
    }
    e->n_records += n;
    e->keys_used += new_words;
    e->finalised = last_batch;

    if (memory == FQD_MEM_HOST) {
        HIP_TRY(e, hipMemcpyAsync(keep, d_keep, n, hipMemcpyDeviceToHost, e->stream));
        return check_state(e);
    }
    return FQD_OK;
}

void* fqd_engine_stream(fqd_engine* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

int fqd_submit(fqd_engine* e, const fqd_reads* seg, uint64_t n, int memory, uint8_t* keep)
{
    return submit_impl(e, seg, n, memory, keep, false);
}

int fqd_submit_final(fqd_engine* e, const fqd_reads* seg, uint64_t n, int memory, uint8_t* keep)
{
    return submit_impl(e, seg, n, memory, keep, true);
}

// An event recorded on `from`, waited for by `to`: a later record of the same event does not move a wait already queued.
static int order_streams(fqd_engine* e, hipStream_t from, hipStream_t to)
{
    if (!e) return FQD_ERR_ARG;
    if (from == to) return FQD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->order_event) HIP_TRY(e, hipEventCreateWithFlags(&e->order_event, hipEventDisableTiming));
    HIP_TRY(e, hipEventRecord(e->order_event, from));
    HIP_TRY(e, hipStreamWaitEvent(to, e->order_event, 0));
    return FQD_OK;
}

int fqd_engine_wait_stream(fqd_engine* e, void* stream) { return e ? order_streams(e, static_cast<hipStream_t>(stream), e->stream) : FQD_ERR_ARG; }
int fqd_stream_wait_engine(fqd_engine* e, void* stream) { return e ? order_streams(e, e->stream, static_cast<hipStream_t>(stream)) : FQD_ERR_ARG; }

int fqd_engine_sync(fqd_engine* e)
{
    if (!e) return FQD_ERR_ARG;
    HIP_TRY(e, hipSetDevice(e->device));
    return check_state(e);
}

int fqd_bad_base(const fqd_engine* e, uint64_t* record, uint32_t* segment, uint32_t* position, uint8_t* byte)
{
    if (!e || !e->has_bad) return FQD_ERR_ARG;
    if (record) *record = e->bad_record;
    if (segment) *segment = e->bad_seg;
    if (position) *position = e->bad_pos;
    if (byte) *byte = e->bad_byte;
    return FQD_OK;
}

int fqd_get_stats(fqd_engine* e, fqd_stats* out)
{
    if (!e || !out) return FQD_ERR_ARG;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(e->h_state, e->d_state, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    out->records = e->n_records; out->duplicates = e->h_state[1];
    out->table_slots = e->slots; out->key_bytes = e->keys_used * sizeof(uint64_t);
    return FQD_OK;
}

int fqd_get_profile(fqd_engine* e, fqd_profile* out)
{
    if (!e || !out) return FQD_ERR_ARG;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    drain_profile(e);
    *out = e->prof;
    return FQD_OK;
}

int fqd_reset_profile(fqd_engine* e)
{
    if (!e) return FQD_ERR_ARG;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    drain_profile(e);
    e->prof = fqd_profile{};
    return FQD_OK;
}

// ---- multi-GPU halves -------------------------------------------------------------

int fqd_encode_uniform(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint64_t* records)
{
    if (!e) return FQD_ERR_ARG;
    if (!seg || (n && !records)) return e->fail(FQD_ERR_ARG, "fqd_encode_uniform: bad arguments");
    for (int s = 0; s < e->S; ++s)
        if (!seg_is_uniform(seg[s])) return e->fail(FQD_ERR_ARG, "fqd_encode_uniform: uniform batches only");
    if (n == 0) return FQD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    SegView sv[2] = {{nullptr, nullptr, nullptr, 0, 0}, {nullptr, nullptr, nullptr, 0, 0}};
    for (int s = 0; s < e->S; ++s) { sv[s].bases = seg[s].bases; sv[s].ulen = seg[s].uniform_len; sv[s].ustride = seg[s].uniform_stride; }
    const uint32_t W = seg_words(seg[0].uniform_len) + (e->S == 2 ? seg_words(seg[1].uniform_len) : 0u);
    KeyStore ks{records, nullptr, W, W + 1, 1};
    // the hashes also go, back to back, to the engine's scratch: fqd_partition_records counts owners
    // from there instead of picking word 0 out of every record
    int rc = reserve(e, e->hashes, n * sizeof(uint64_t));
    if (rc) return rc;
    rc = launch_encode(e, sv, true, seg, n, 0, ks, e->hashes.as<uint64_t>());
    e->hashed_records = rc == FQD_OK ? records : nullptr; e->hashed_n = n; e->hashed_rec_words = W + 1;
    return rc;
}

uint32_t fqd_padded_key_words(uint32_t max_len0, uint32_t max_len1) { return 1u + seg_words(max_len0) + seg_words(max_len1); }

int fqd_encode_padded(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t max_len0, uint32_t max_len1, uint64_t* records)
{
    if (!e) return FQD_ERR_ARG;
    if (!seg || (n && !records) || max_len0 == 0 || (e->S == 1 && max_len1 != 0)) return e->fail(FQD_ERR_ARG, "fqd_encode_padded: bad arguments");
    if (n == 0) return FQD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    bool uniform = true;
    SegView sv[2] = {{nullptr, nullptr, nullptr, 0, 0}, {nullptr, nullptr, nullptr, 0, 0}};
    for (int s = 0; s < e->S; ++s) {
        if (!seg_is_uniform(seg[s])) { if (!seg[s].offsets || !seg[s].lengths) return e->fail(FQD_ERR_ARG, "fqd_encode_padded: offsets and lengths go together"); uniform = false; }
        sv[s].bases = seg[s].bases; sv[s].offsets = seg[s].offsets; sv[s].lengths = seg[s].lengths;
        sv[s].ulen = seg[s].uniform_len; sv[s].ustride = seg[s].uniform_stride; sv[s].clamp = s ? max_len1 : max_len0;
    }
    const uint32_t K = fqd_padded_key_words(max_len0, e->S == 2 ? max_len1 : 0u), stride = K + 1u;
    int rc = reserve(e, e->hashes, n * sizeof(uint64_t));
    if (rc) return rc;
    if ((rc = reserve(e, e->pad_koff, n * sizeof(uint64_t)))) return rc;
    {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(records, 0, n * uint64_t(stride) * sizeof(uint64_t), e->stream));      // what a shorter read leaves of its slot is zeros
        hipLaunchKernelGGL(padded_slots_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream, sv[0], sv[1], uint32_t(e->S == 2), n,
                           max_len0, e->S == 2 ? max_len1 : 0u, stride, e->pad_koff.as<uint64_t>(), reinterpret_cast<unsigned long long*>(e->d_state + 4));
    }
    KeyStore ks{records, e->pad_koff.as<uint64_t>(), 0, 0, 1};
    fqd_reads padded[2] = {seg[0], seg[1]};
    rc = launch_encode(e, sv, uniform, padded, n, 0, ks, e->hashes.as<uint64_t>());
    e->hashed_records = rc == FQD_OK ? records : nullptr; e->hashed_n = n; e->hashed_rec_words = stride;
    return rc;
}

static int partition_impl(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words,
                          uint32_t n_parts, uint64_t* out, uint64_t* counts, uint32_t* origin, uint32_t strip, uint64_t slab_cap = 0);

int fqd_partition_keys(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words,
                       uint32_t n_parts, uint64_t* out_keys, uint64_t* counts, uint32_t* origin)
{
    if (e && key_words == 0) return e->fail(FQD_ERR_ARG, "fqd_partition_keys: records without key words");
    return partition_impl(e, records, n, key_words, n_parts, out_keys, counts, origin, 1u);
}

int fqd_partition_slabs(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words,
                        uint32_t n_parts, uint64_t slab_cap, uint64_t* out_keys, uint64_t* counts, uint32_t* origin)
{
    if (e && (key_words == 0 || slab_cap == 0)) return e->fail(FQD_ERR_ARG, "fqd_partition_slabs: bad arguments");
    return partition_impl(e, records, n, key_words, n_parts, out_keys, counts, origin, 1u, slab_cap);
}

// Encode + group by owner into slabs cut into one sub-slab per chunk of the input (fqd_kernels.hpp, encode_chunks).
// One pass where it applies — reads of one fixed length per mate, equally spaced, a tile of which fits LDS with room
// for its keys, at most kGroupParts owners, chunks of whole tiles — otherwise, or with FQD_SLABS_EXACT, the three-step
// path through an internal record buffer: slabs filled from their first slot on (the same thing seen as full, partial
// and empty sub-slabs), spill region written.  chunk_counts[p * n_chunks + c] and totals[p] are the TRUE counts either
// way; the one-pass form does not write a key whose sub-slab is full.
int fqd_encode_slabs(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t n_parts, uint64_t chunk_reads, uint32_t n_chunks, uint64_t sub_cap,
                     uint64_t* out_keys, uint64_t* chunk_counts, uint64_t* totals, uint32_t* origin, uint32_t flags)
{
    return fqd_encode_slabs_hashed(e, seg, n, n_parts, chunk_reads, n_chunks, sub_cap, out_keys, nullptr, chunk_counts, totals, origin, flags);
}

int fqd_encode_slabs_hashed(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t n_parts, uint64_t chunk_reads, uint32_t n_chunks, uint64_t sub_cap,
                            uint64_t* out_keys, uint64_t* out_hashes, uint64_t* chunk_counts, uint64_t* totals, uint32_t* origin, uint32_t flags)
{
    if (!e) return FQD_ERR_ARG;
    const uint64_t slab_cap = uint64_t(n_chunks) * sub_cap;
    if (!seg || !n_parts || !n_chunks || !sub_cap || !chunk_reads || !chunk_counts || !totals || (n && (!out_keys || !origin)) || n > 0xFFFFFFFFull ||
        slab_cap * n_parts + n > 0xFFFFFFFFull || n > chunk_reads * n_chunks)
        return e->fail(FQD_ERR_ARG, "fqd_encode_slabs: bad arguments");
    for (int s = 0; s < e->S; ++s)
        if (!seg_is_uniform(seg[s])) return e->fail(FQD_ERR_ARG, "fqd_encode_slabs: uniform batches only (fqd_encode_padded + fqd_partition_slabs take the others)");
    HIP_TRY(e, hipSetDevice(e->device));
    const uint32_t W = seg_words(seg[0].uniform_len) + (e->S == 2 ? seg_words(seg[1].uniform_len) : 0u);
    const char* group_env = std::getenv("FQD_ENCODE_GROUP");                 // FQD_ENCODE_GROUP=0: always the three steps (read per call: tests switch it inside one process)
    const bool one_pass_on = !group_env || std::atoi(group_env) != 0;
    // a tile: 256 reads, or 128 pairs (one lane per mate); every key is parked over its own read's bytes in LDS (encode_staged)
    const uint32_t per_tile = e->S == 2 ? kBlock / 2 : kBlock;
    const uint64_t tile0 = (uint64_t(per_tile) * seg[0].uniform_stride + 32 + 15) & ~15ull;
    const uint64_t tile1 = e->S == 2 ? ((uint64_t(per_tile) * seg[1].uniform_stride + 32 + 15) & ~15ull) : 0;
    const uint64_t tile_bytes = tile0 + tile1;
    const uint32_t W_0 = seg_words(seg[0].uniform_len), W_1 = e->S == 2 ? seg_words(seg[1].uniform_len) : 0u;
    bool one_pass = one_pass_on && !(flags & FQD_SLABS_EXACT) && n_parts <= kGroupParts && !(e->flags & FQD_FLAG_NO_STAGE) && chunk_reads % per_tile == 0 &&
                    W > 1 && tile_bytes <= 64 * 1024 && seg[0].uniform_stride >= 8u * W_0 + 15u && seg[0].uniform_len > 0;
    if (e->S == 2) one_pass = one_pass && seg[1].uniform_stride >= 8u * W_1 + 15u && seg[1].uniform_len > 0;
    if (!one_pass) {
        int rc = reserve(e, e->slab_records, std::max<uint64_t>(n, 1) * uint64_t(W + 1) * sizeof(uint64_t));
        if (rc) return rc;
        if ((rc = fqd_encode_uniform(e, seg, n, e->slab_records.as<uint64_t>()))) return rc;
        if ((rc = fqd_partition_slabs(e, e->slab_records.as<uint64_t>(), n, W, n_parts, slab_cap, out_keys, totals, origin))) return rc;
        HIP_TRY(e, hipMemsetAsync(totals + n_parts, 0, sizeof(uint64_t), e->stream));
        HIP_TRY(e, hipMemsetAsync(totals + n_parts, 1, 1, e->stream));        // layout word: 1 = slabs filled from their first slot on
        const uint64_t cells = uint64_t(n_parts) * n_chunks;
        hipLaunchKernelGGL(classic_chunk_counts_kernel, dim3(uint32_t((cells + 255) / 256)), dim3(256), 0, e->stream,
                           static_cast<const uint64_t*>(totals), n_parts, n_chunks, sub_cap, chunk_counts);
        HIP_TRY(e, hipGetLastError());
        return FQD_OK;
    }
    {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(origin, 0xFF, slab_cap * n_parts * sizeof(uint32_t), e->stream));
        HIP_TRY(e, hipMemsetAsync(chunk_counts, 0, uint64_t(n_parts) * n_chunks * sizeof(uint64_t), e->stream));
        HIP_TRY(e, hipMemsetAsync(totals, 0, (n_parts + 1) * sizeof(uint64_t), e->stream));       // (and the layout word: 0 = sub-slab by sub-slab)
        if (n == 0) return FQD_OK;
    }
    int rc = reserve(e, e->part_scratch, 256);
    if (rc) return rc;
    uint32_t* next_chunk = e->part_scratch.as<uint32_t>();
    {
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(next_chunk, 0, 256, e->stream));
    }
    SegView sv{seg[0].bases, nullptr, nullptr, seg[0].uniform_len, seg[0].uniform_stride};
    const uint64_t hash_and = (e->flags & FQD_FLAG_WEAK_HASH) ? 0x00000000FFFFFFC0ull : ~0ull;
    const uint32_t magic = uint32_t(((1ull << 32) + W - 1) / W);
    const uint32_t chunk_tiles = uint32_t(chunk_reads / per_tile);
    const uint32_t used_chunks = uint32_t((n + chunk_reads - 1) / chunk_reads);
    const uint32_t grid = uint32_t(std::min<uint64_t>(used_chunks, uint64_t(e->n_cu) * 4u));
    Bracket br(e, K_ENCODE, n);
    if (e->S == 1)
        hipLaunchKernelGGL(encode_chunks_kernel, dim3(grid), dim3(kBlock), size_t(tile_bytes), e->stream, sv, n, W, n_parts, chunk_tiles, n_chunks, used_chunks, sub_cap,
                           out_keys, origin, chunk_counts, reinterpret_cast<unsigned long long*>(totals), next_chunk, e->d_state, magic, hash_and, out_hashes);
    else {
        SegView sv1{seg[1].bases, nullptr, nullptr, seg[1].uniform_len, seg[1].uniform_stride};
        hipLaunchKernelGGL(encode_chunks_pe_kernel, dim3(grid), dim3(kBlock), size_t(tile_bytes), e->stream, sv, sv1, n, W_0, W_1, n_parts, chunk_tiles, n_chunks, used_chunks, sub_cap,
                           out_keys, origin, chunk_counts, reinterpret_cast<unsigned long long*>(totals), next_chunk, e->d_state, uint32_t(tile0), magic, hash_and, out_hashes);
    }
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

static int partition_impl(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words,
                          uint32_t n_parts, uint64_t* out, uint64_t* counts, uint32_t* origin, uint32_t strip, uint64_t slab_cap)
{
    if (!e) return FQD_ERR_ARG;
    if (!n_parts || n_parts > 1024 || !counts || (n && (!records || !out || !origin)) || n > 0xFFFFFFFFull ||
        (slab_cap && (!origin || slab_cap * n_parts + n > 0xFFFFFFFFull)))
        return e->fail(FQD_ERR_ARG, "fqd_partition_records: bad arguments");
    HIP_TRY(e, hipSetDevice(e->device));
    if (slab_cap) {                                          // slab slots no record reaches say so
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemsetAsync(origin, 0xFF, slab_cap * n_parts * sizeof(uint32_t), e->stream));
    }
    if (n == 0) { HIP_TRY(e, hipMemsetAsync(counts, 0, n_parts * sizeof(uint64_t), e->stream)); return FQD_OK; }
    const uint32_t n_blocks = uint32_t((n + kBlock - 1) / kBlock);
    const uint32_t rec_words = key_words + 1;
    const uint64_t cells = uint64_t(n_parts) * n_blocks;
    int rc = reserve(e, e->part_scratch, (cells + 2 * uint64_t(n_parts) + 2) * sizeof(uint64_t));
    if (rc) return rc;
    uint64_t* c2 = e->part_scratch.as<uint64_t>();
    ulonglong2* slab_of = reinterpret_cast<ulonglong2*>(c2 + ((cells + 1) & ~1ull));
    Bracket br(e, K_OTHER, 0);
    const bool compact = records == e->hashed_records && n == e->hashed_n && rec_words == e->hashed_rec_words;
    hipLaunchKernelGGL(part_count_kernel, dim3(n_blocks), dim3(kBlock), n_parts * sizeof(uint32_t), e->stream,
                       compact ? e->hashes.as<uint64_t>() : records, compact ? 1u : rec_words, n, n_parts, c2, n_blocks);
    if ((rc = scan_exclusive(e, c2, cells, 0, nullptr))) return rc;
    const size_t scatter_lds = size_t(kBlock) * rec_words * sizeof(uint64_t) + kBlock * sizeof(uint64_t) + size_t(n_parts) * 4 * sizeof(uint32_t);
    if (scatter_lds > 64 * 1024) return e->fail(FQD_ERR_ARG, "fqd_partition_records: records too long for the staged partition");
    const uint32_t rw_magic = rec_words > 1 ? uint32_t(((1ull << 32) + rec_words - 1) / rec_words) : 0xFFFFFFFFu;
    if (slab_cap)
        hipLaunchKernelGGL(part_slab_offsets_kernel, dim3(1), dim3(64), 0, e->stream,
                           static_cast<const uint64_t*>(c2), n_parts, n_blocks, n, slab_cap, slab_of);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(n_blocks), dim3(kBlock), scatter_lds, e->stream,
                       records, n, rec_words, n_parts, static_cast<const uint64_t*>(c2), n_blocks, out, origin, rw_magic, strip,
                       slab_cap, static_cast<const ulonglong2*>(slab_of));
    hipLaunchKernelGGL(part_totals_kernel, dim3((n_parts + 63) / 64), dim3(64), 0, e->stream,
                       static_cast<const uint64_t*>(c2), n_parts, n_blocks, n, counts);
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

// Keys that arrive without their hash (the sharded exchange): the owner is an ordinary uniform
// engine, the keys lie back to back at the tail of its key store.
static int prepare_keys(fqd_engine* e, uint64_t n, uint32_t len0, uint32_t len1)
{
    const bool opaque = len1 == FQD_OPAQUE_KEYS;
    if (e->S == 1 && len1 != 0 && !opaque) return e->fail(FQD_ERR_ARG, "keys: single-end engine given a mate-2 length");
    if (e->n_records + n > 0xFFFFFFFEull) return e->fail(FQD_ERR_CAPACITY, "more than 2^32-2 records in one engine");
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->have_shape) {
        e->have_shape = true; e->ragged = false;
        e->L0 = len0; e->L1 = len1; e->W0 = opaque ? len0 : seg_words(len0) + seg_words(len1);
    } else if (e->ragged || e->L0 != len0 || e->L1 != len1)
        return e->fail(FQD_ERR_ARG, "keys: engine holds keys of another shape or layout");
    if (e->W0 == 0) return e->fail(FQD_ERR_ARG, "keys: empty keys");
    int rc;
    if ((rc = ensure_table(e, e->n_records + n))) return rc;
    const uint64_t need = e->keys_used + n * uint64_t(e->W0);
    return reserve(e, e->keys, std::max<uint64_t>(need, 64) * sizeof(uint64_t), e->keys_used * sizeof(uint64_t));
}

int fqd_reserve_keys(fqd_engine* e, uint64_t n, uint32_t len0, uint32_t len1, uint64_t** slot)
{
    if (!e) return FQD_ERR_ARG;
    if (!slot) return e->fail(FQD_ERR_ARG, "fqd_reserve_keys: bad arguments");
    const int rc = prepare_keys(e, n, len0, len1);
    if (rc) return rc;
    *slot = e->keys.as<uint64_t>() + e->keys_used;
    return FQD_OK;
}

// The owner side of a key shape that changes in mid-run (a multi-GPU run that meets a read longer than any before,
// or reads of several lengths after blocks of one length): every key the engine holds is laid out again as an opaque
// key of `new_words` words — [len0 | len1 << 32][words][zeros] for keys of known mate lengths, [words][zeros] for
// keys that are opaque already — which is exactly what fqd_encode_padded makes of the same read under the wider
// maxima, so old and new keys stay comparable word for word.  Record numbers do not change; the set is rebuilt from
// the new keys (their placement hash runs over all words).
int fqd_widen_keys(fqd_engine* e, uint32_t new_words)
{
    if (!e) return FQD_ERR_ARG;
    if (new_words == 0) return e->fail(FQD_ERR_ARG, "fqd_widen_keys: empty keys");
    if (e->finalised) return e->fail(FQD_ERR_ARG, kFinalised);
    if (e->ragged) return e->fail(FQD_ERR_ARG, "fqd_widen_keys: this engine holds keys of several lengths in the ragged layout (fqd_submit); only uniform key stores are widened");
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->have_shape || e->n_records == 0) {
        e->have_shape = true; e->ragged = false; e->L0 = new_words; e->L1 = FQD_OPAQUE_KEYS; e->W0 = new_words; e->keys_used = 0;
        return FQD_OK;
    }
    const bool opaque = e->L1 == FQD_OPAQUE_KEYS;
    const uint32_t lead = opaque ? 0u : 1u;
    if (new_words < e->W0 + lead) return e->fail(FQD_ERR_ARG, "fqd_widen_keys: the new keys are narrower than the ones held");
    if (opaque && new_words == e->W0) return FQD_OK;
    int rc;
    DevBuf nk;
    const uint64_t words = e->n_records * uint64_t(new_words);
    if ((rc = reserve(e, nk, std::max<uint64_t>(words + words / 2, 1024) * sizeof(uint64_t)))) return rc;
    void* nt = nullptr;
    const bool have_table = e->table.p && !e->table_clear && !e->table_stale;
    if (have_table) {
        const hipError_t err = hipMalloc(&nt, e->slots * sizeof(uint64_t));
        if (err != hipSuccess) { release(nk); return e->fail_hip("hipMalloc(table)", err); }
    }
    {
        Bracket br(e, K_OTHER, 0);
        const uint64_t header = uint64_t(e->L0) | (uint64_t(e->L1) << 32);
        hipLaunchKernelGGL(widen_keys_kernel, dim3(grid_for(e, words)), dim3(kBlock), 0, e->stream,
                           e->keys.as<uint64_t>(), nk.as<uint64_t>(), e->n_records, e->W0, new_words, lead, header);
        if (have_table) {
            HIP_TRY(e, hipMemsetAsync(nt, 0xFF, e->slots * sizeof(uint64_t), e->stream));
            const KeyStore ks{nk.as<uint64_t>(), nullptr, new_words, new_words, 0};
            hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(e, e->slots)), dim3(kBlock), 0, e->stream,
                               e->table.as<uint64_t>(), e->slots, static_cast<uint64_t*>(nt), e->slots - 1,
                               (1ull << e->seg_bits) - 1, ks, new_words, FQD_OPAQUE_KEYS, 0u,
                               !(e->flags & FQD_FLAG_WEAK_HASH) ? ~0ull : 0x00000000FFFFFFC0ull,
                               e->tag_mask, reinterpret_cast<unsigned long long*>(e->d_state + 1));
        }
        HIP_TRY(e, hipGetLastError());
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    release(e->keys);
    e->keys = nk; e->keys_used = words;
    if (have_table) { HIP_TRY(e, hipFree(e->table.p)); e->table.p = nt; }
    e->L0 = new_words; e->L1 = FQD_OPAQUE_KEYS; e->W0 = new_words;
    e->hashed_records = nullptr;
    return FQD_OK;
}

static int insert_keys_impl(fqd_engine* e, const uint64_t* keys, uint64_t n, uint32_t len0, uint32_t len1, uint8_t* keep,
                            uint64_t slab_cap, const uint64_t* slab_count, uint64_t* hashes = nullptr);

int fqd_insert_slabs_hashed(fqd_engine* e, const uint64_t* keys, uint64_t* hashes, uint32_t n_slabs, uint64_t slab_cap, const uint64_t* slab_count,
                            uint32_t len0, uint32_t len1, uint8_t* keep)
{
    if (e && (!slab_cap || !slab_count || !hashes)) return e->fail(FQD_ERR_ARG, "fqd_insert_slabs_hashed: bad arguments");
    if (e && len1 == FQD_OPAQUE_KEYS) return e->fail(FQD_ERR_ARG, "fqd_insert_slabs_hashed: opaque keys are hashed by their owner (the source's hash runs over the read, the owner's over the padded key)");
    return insert_keys_impl(e, keys, uint64_t(n_slabs) * slab_cap, len0, len1, keep, slab_cap, slab_count, hashes);
}

int fqd_insert_keys(fqd_engine* e, const uint64_t* keys, uint64_t n, uint32_t len0, uint32_t len1, uint8_t* keep)
{
    return insert_keys_impl(e, keys, n, len0, len1, keep, 0, nullptr);
}

int fqd_insert_slabs(fqd_engine* e, const uint64_t* keys, uint32_t n_slabs, uint64_t slab_cap, const uint64_t* slab_count,
                     uint32_t len0, uint32_t len1, uint8_t* keep)
{
    if (e && (!slab_cap || !slab_count)) return e->fail(FQD_ERR_ARG, "fqd_insert_slabs: bad arguments");
    return insert_keys_impl(e, keys, uint64_t(n_slabs) * slab_cap, len0, len1, keep, slab_cap, slab_count);
}

static int insert_keys_impl(fqd_engine* e, const uint64_t* keys, uint64_t n, uint32_t len0, uint32_t len1, uint8_t* keep,
                            uint64_t slab_cap, const uint64_t* slab_count, uint64_t* given_hashes)
{
    if (!e) return FQD_ERR_ARG;
    if (n && (!keys || !keep)) return e->fail(FQD_ERR_ARG, "fqd_insert_keys: bad arguments");
    if (e->finalised) return e->fail(FQD_ERR_ARG, kFinalised);
    if (n == 0) return FQD_OK;
    int rc;
    if ((rc = prepare_keys(e, n, len0, len1))) return rc;
    const uint64_t first = e->n_records;
    const uint64_t words = n * uint64_t(e->W0);
    uint64_t* tail = e->keys.as<uint64_t>() + e->keys_used;
    if (keys != tail) {                                      // not received in place: one contiguous copy
        Bracket br(e, K_OTHER, 0);
        HIP_TRY(e, hipMemcpyAsync(tail, keys, words * sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream));
    }
    e->hashed_records = nullptr;                              // the hash scratch is reused
    const uint64_t* hashes = given_hashes;
    if (given_hashes) {
        // the hashes came with the keys (the source's encoder had them anyway): only the slots without a key get their word
        Bracket br(e, K_OTHER, 0);
        hipLaunchKernelGGL(mask_unused_hashes_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream, given_hashes, n, slab_cap, slab_count);
    } else {
        if ((rc = reserve(e, e->hashes, n * sizeof(uint64_t)))) return rc;
        hashes = e->hashes.as<uint64_t>();
        Bracket br(e, K_OTHER, 0);
        const uint64_t hash_and = (e->flags & FQD_FLAG_WEAK_HASH) ? 0x00000000FFFFFFC0ull : ~0ull;
        hipLaunchKernelGGL(hash_keys_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream,
                           static_cast<const uint64_t*>(tail), e->W0, n, e->L0, e->L1, uint32_t(e->S == 2), hash_and, e->hashes.as<uint64_t>(),
                           slab_cap, slab_count);
    }
    const KeyStore ks = key_store(e);
    BulkPlan plan;
    if (bulk_applies(e, n) && (rc = bulk_plan(e, n, plan))) return rc;
    if (plan.ok) { if ((rc = launch_bulk_insert(e, ks, hashes, 1, n, first, keep, plan, false))) return rc; }
    else if ((rc = launch_insert(e, ks, hashes, 1, n, first, keep))) return rc;
    e->n_records += n; e->keys_used += words;
    return FQD_OK;
}

int fqd_scatter_flags(fqd_engine* e, const uint8_t* flags, const uint32_t* origin, uint64_t n, uint8_t* keep_out)
{
    if (!e) return FQD_ERR_ARG;
    if (n && (!flags || !origin || !keep_out)) return e->fail(FQD_ERR_ARG, "fqd_scatter_flags: bad arguments");
    if (n == 0) return FQD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    Bracket br(e, K_OTHER, 0);
    hipLaunchKernelGGL(scatter_flags_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream, flags, origin, n, keep_out);
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_synth_reads(fqd_engine* e, uint64_t seed, uint64_t first, uint64_t n, uint32_t len,
                    uint32_t dup_permille, int mate, uint8_t* bases, uint8_t* expect_keep)
{
    if (!e) return FQD_ERR_ARG;
    if ((n && len && !bases) || dup_permille > 1000 || (mate != 0 && mate != 1))
        return e->fail(FQD_ERR_ARG, "fqd_synth_reads: bad arguments");
    if (n == 0) return FQD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(e, n)), dim3(kBlock), 0, e->stream,
                       seed, first, n, len, dup_permille, mate, bases, expect_keep);
    HIP_TRY(e, hipGetLastError());
    return FQD_OK;
}

} // extern "C"
