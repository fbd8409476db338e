// run_multi.cpp — the runs over several GPUs (FQD_DEVICES): ordered single-end / paired through the shard group, and
// `--unordered` by tag ranges.
#include "run_common.hpp"

namespace fqdhost {
using namespace detail;

// ---------------------------------------------------------------------------
// The same ordered runs over several GPUs (FQD_DEVICES): one process, one engine per GPU, driven through the shard
// group of include/fqdupaway.h (fqd_shard_*; csrc/fqd_shard.hip) — the same code bench.py measures with one process
// per GPU.  A round deals the next batches to the ranks in file order (global order = round, rank, position), the
// group moves the keys to their owners in fixed-size slabs over RCCL (FQD_EXCHANGE=copy: peer copies) and brings the
// flags back; rounds are pipelined, so the writers get round k-1 while round k is on the GPUs.  What travels are
// fixed-size keys: of exactly the reads' length when the first round's reads all have one (per mate), otherwise
// (trimmed reads) padded to the longest read of the first round, rounded up (include/fqdupaway.h, fqd_encode_padded).
void HashDupRemover::run_ordered_multi(int S, const std::string* in, const std::string* out)
{
    const std::vector<int>& devs = tuning_.devices;
    const int N = static_cast<int>(devs.size());
    std::unique_ptr<OutputFile> sink[2];
    for (int s = 0; s < S; ++s) sink[s] = std::make_unique<OutputFile>(out[s]);

    Side side[2];
    for (int s = 0; s < S; ++s) {
        side[s].open_file(in[s], format_, S == 2, tuning_.block_bytes);
        HIP_OK(hipSetDevice(devs[0]));
        side[s].prime(2 * N + 2, devs[0]);
        if (side[s].available() == 0 && side[s].failed && !side[s].held_back) {
            std::cerr << side[s].failure.diag;
            throw std::runtime_error(side[s].failure.what);
        }
    }

    // one rank per listed GPU: stream, engine, its own pool of batches (dealt, awaiting flags, being written)
    struct Rank {
        int device = 0; hipStream_t stream = nullptr; std::unique_ptr<EngineHandle> eng;
        Channel<Work> pool; std::vector<std::unique_ptr<Work>> works;
        ~Rank() { eng.reset(); if (stream) { (void)hipSetDevice(device); (void)hipStreamDestroy(stream); } }
    };
    std::vector<std::unique_ptr<Rank>> rank;
    for (int r = 0; r < N; ++r) {
        rank.emplace_back(new Rank());
        Rank& k = *rank.back();
        k.device = devs[r];
        HIP_OK(hipSetDevice(k.device));
        HIP_OK(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
        k.eng = std::make_unique<EngineHandle>(S, k.device, k.stream);
        for (int w = 0; w < 4; ++w) { k.works.emplace_back(new Work()); k.works.back()->S = S; k.works.back()->home = &k.pool; k.pool.push(k.works.back().get()); }
    }
    struct ShardGuard { fqd_shard* g = nullptr; ~ShardGuard() { if (g) fqd_shard_destroy(g); } } shard;

    Channel<Work> spare;                                       // only the stop marker lives here
    Work stop_marker; stop_marker.S = S; stop_marker.home = &spare;
    SurvivorWriters writers(S, sink, &spare);

    uint64_t next_index = 0, total_dups = 0;
    bool bad_base = false; uint8_t bad_byte = 0; uint64_t bad_at = ~0ull;
    // the group's key shape is fixed by the first round: one length per mate everywhere -> fixed-size keys of exactly that
    // length; anything else (trimmed reads) -> keys padded to the longest read seen there, rounded up (FQD_SHARD_PADDED)
    uint32_t len0 = 0, len1 = 0; bool have_shape_of[2] = {false, false}; bool all_uniform = true, padded = false;
    constexpr size_t kMaxBatch = 8u << 20;
    size_t round_reads = kMaxBatch;                            // most records a rank brings to a round: fixed when the group is made
    std::vector<Work*> round, previous;                        // this round's batch per rank (null: none), last round's
    uint64_t rounds_started = 0;
    auto shard_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(shard.g)); };

    // flags of a finished round: to the host, then to the writers, in rank order = file order
    auto deliver = [&](std::vector<Work*>& batch, uint64_t round_no) {
        const int rc = fqd_shard_wait(shard.g, round_no);
        if (rc == FQD_ERR_BAD_BASE) {
            // the engines remember their first bad byte for good: only the first round that reports one places the cut
            int32_t lr = 0; uint64_t rec = 0; uint32_t sg = 0, pos = 0; uint8_t byte = 0;
            if (!bad_base && fqd_shard_bad_base(shard.g, round_no, &lr, &rec, &sg, &pos, &byte) == FQD_OK && batch[size_t(lr)]) {
                bad_base = true; bad_byte = byte; bad_at = batch[size_t(lr)]->first_index + rec;
            }
        } else shard_ok(rc);
        for (int r = 0; r < N; ++r) {
            Work* w = batch[size_t(r)];
            if (!w) continue;
            HIP_OK(hipSetDevice(rank[size_t(r)]->device));
            HIP_OK(hipMemcpy(w->keep.p, w->d_keep.p, w->n, hipMemcpyDeviceToHost));
            if (bad_base) w->emit_below = bad_at;              // flags before the bad record are valid, nothing from it on is written
            writers.hand_over(w);
            batch[size_t(r)] = nullptr;
        }
    };

    // The group is made for one key shape; the reference keys a read of any length wherever it stands in the file
    // (seq_utils.cpp:35-49).  A batch that does not fit the shape — reads of several lengths after blocks of one length,
    // or a read longer than the padded width — retires the group: its rounds are finished and handed to the writers,
    // every owner lays the keys it holds out again at the new width (fqd_widen_keys) and a new group is made for it.
    bool first_group = true, widen = false;
    auto retire_group = [&]() {
        if (rounds_started) {
            const int rc = fqd_shard_flush(shard.g);
            if (rc != FQD_OK && rc != FQD_ERR_BAD_BASE) shard_ok(rc);
            if (!previous.empty()) deliver(previous, rounds_started - 1);
        }
        fqd_shard_destroy(shard.g); shard.g = nullptr;
        rounds_started = 0; previous.clear();
        widen = true; all_uniform = false;
    };

    try {
        while (!bad_base) {
            // ---- deal the next batches to the ranks, in file order ----------------------------------
            round.assign(size_t(N), nullptr);
            std::vector<fqd_reads> seg(size_t(N) * size_t(S));
            std::vector<uint64_t> n_of(size_t(N), 0);
            std::vector<uint8_t*> keep_of(size_t(N), nullptr);
            bool any = false;
            for (int r = 0; r < N && !bad_base; ++r) {
                size_t n = round_reads;
                for (int s = 0; s < S; ++s) n = std::min(n, side[s].available());
                if (n == 0) break;
                // the batch's shape first: does the group take it?
                bool uniform_of[2] = {true, true}; uint32_t longest_of[2] = {0, 0}; uint64_t stride_of[2] = {0, 0};
                bool fits = true;
                for (int s = 0; s < S; ++s) {
                    const RecordRef* rr = &side[s].cur->recs[side[s].pos];
                    const uint64_t stride = n > 1 ? rr[1].seq_start() - rr[0].seq_start() : rr[0].size;
                    bool uniform = stride <= 0xFFFFFFFFull && rr[0].seq_len > 0;
                    uint32_t longest = rr[0].seq_len;
                    for (size_t i = 1; i < n; ++i) {
                        uniform = uniform && rr[i].seq_len == rr[0].seq_len && rr[i].seq_start() - rr[i - 1].seq_start() == stride;
                        longest = std::max(longest, rr[i].seq_len);
                    }
                    uniform_of[s] = uniform; longest_of[s] = longest; stride_of[s] = stride;
                    // what the group was made for: one fixed length per mate, or (padded keys) anything up to a maximum
                    if (shard.g && !padded && (!uniform || rr[0].seq_len != (s ? len1 : len0))) fits = false;
                    if (shard.g && padded && longest > (s ? len1 : len0)) fits = false;
                }
                if (!fits) {
                    retire_group();
                    if (bad_base) break;                          // a round the group still owed reported an unknown base: the output ends there
                }
                Rank& k = *rank[size_t(r)];
                Work* w = k.pool.pop();
                w->stop = false; w->n = n; w->first_index = next_index; w->emit_below = ~0ull;
                HIP_OK(hipSetDevice(k.device));
                w->keep.reserve(n); w->d_keep.reserve(n);
                for (int s = 0; s < S; ++s) {
                    PooledBlock* b = side[s].cur;
                    b->acquire();
                    w->blk[s] = b; w->begin[s] = side[s].pos;
                    const RecordRef* rr = &b->recs[side[s].pos];
                    const uint64_t text_lo = rr[0].start, text_hi = rr[n - 1].start + rr[n - 1].size;
                    const uint64_t stride = stride_of[s];
                    const bool uniform = uniform_of[s];
                    if (!shard.g) {
                        all_uniform = all_uniform && uniform && (!have_shape_of[s] || rr[0].seq_len == (s ? len1 : len0));
                        have_shape_of[s] = true;
                        (s ? len1 : len0) = std::max(s ? len1 : len0, longest_of[s]);
                    }
                    w->d_text[s].reserve(text_hi - text_lo + 32);
                    HIP_OK(hipMemcpyAsync(w->d_text[s].p, b->text.p + text_lo, text_hi - text_lo, hipMemcpyHostToDevice, k.stream));
                    fqd_reads& d = seg[size_t(r) * size_t(S) + size_t(s)];
                    d = fqd_reads{};
                    if (uniform) {
                        d.bases = reinterpret_cast<const uint8_t*>(w->d_text[s].p) + (rr[0].seq_start() - text_lo);
                        d.uniform_len = rr[0].seq_len; d.uniform_stride = static_cast<uint32_t>(stride);
                    } else {
                        w->off[s].reserve(n); w->len[s].reserve(n); w->d_off[s].reserve(n); w->d_len[s].reserve(n);
                        for (size_t i = 0; i < n; ++i) { w->off[s].p[i] = rr[i].seq_start() - text_lo; w->len[s].p[i] = rr[i].seq_len; }
                        HIP_OK(hipMemcpyAsync(w->d_off[s].p, w->off[s].p, n * sizeof(uint64_t), hipMemcpyHostToDevice, k.stream));
                        HIP_OK(hipMemcpyAsync(w->d_len[s].p, w->len[s].p, n * sizeof(uint32_t), hipMemcpyHostToDevice, k.stream));
                        d.bases = reinterpret_cast<const uint8_t*>(w->d_text[s].p);
                        d.offsets = w->d_off[s].p; d.lengths = w->d_len[s].p;
                    }
                    side[s].pos += n;
                }
                round[size_t(r)] = w; n_of[size_t(r)] = n; keep_of[size_t(r)] = w->d_keep.p;
                next_index += n;
                any = true;
            }
            if (bad_base) {
                // found while the group was being retired: what this round had dealt is never keyed, nor written
                for (Work*& w : round) if (w) { next_index -= w->n; for (int s = 0; s < S; ++s) if (w->blk[s]) w->blk[s]->release(); w->home->push(w); w = nullptr; }
                break;
            }
            if (!any) break;
            if (!shard.g) {
                std::vector<fqd_engine*> engines;
                for (auto& k : rank) engines.push_back(k->eng->e);
                uint8_t id[FQD_SHARD_ID_BYTES] = {};
                fqd_shard_config cfg{};
                cfg.world = N; cfg.n_local = N; cfg.first_rank = 0;
                cfg.transport = tuning_.use_rccl ? FQD_SHARD_RCCL : FQD_SHARD_COPY;
                // a batch is what one input block holds: size the group's buffers by the first round's batches with room to
                // spare (later batches are cut to that) instead of by the 8 Mi-record ceiling
                if (first_group) {
                    size_t most = 0;
                    for (uint64_t x : n_of) most = std::max<size_t>(most, size_t(x));
                    round_reads = std::min(kMaxBatch, most + most / 4 + 1024);
                }
                const char* force = std::getenv("FQD_SHARD_PADDED");
                padded = !all_uniform || (force && force[0] == '1');
                if (padded) {
                    // room to spare above the longest read met so far: a whole 32-base group costs one key word
                    len0 = (len0 + 31u) / 32u * 32u; len1 = (len1 + 31u) / 32u * 32u;
                    if (const char* v = std::getenv("FQD_SHARD_MAX_LEN")) { const uint32_t x = uint32_t(std::strtoul(v, nullptr, 10)); len0 = std::max(len0, x); if (S == 2) len1 = std::max(len1, x); }
                    cfg.flags |= FQD_SHARD_PADDED;
                }
                if (widen) {
                    StageClock::Scope t("multi: keys laid out again for a wider shape");
                    for (auto& k : rank) {
                        HIP_OK(hipSetDevice(k->device));
                        if (fqd_widen_keys(k->eng->e, fqd_padded_key_words(len0, S == 2 ? len1 : 0u)) != FQD_OK)
                            throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(k->eng->e));
                    }
                    widen = false;
                }
                // FQD_SHARD_SEND_HASH=1: every key's placement hash travels with it (12.5 % more bytes on the links, no re-hash
                // pass at the owners); the library ignores it for padded keys
                if (const char* v = std::getenv("FQD_SHARD_SEND_HASH")) if (std::atoi(v) != 0) cfg.flags |= FQD_SHARD_SEND_HASH;
                cfg.round_reads = round_reads; cfg.len0 = len0; cfg.len1 = S == 2 ? len1 : 0;
                if (const char* v = std::getenv("FQD_SHARD_SLAB")) cfg.slab_records = std::strtoull(v, nullptr, 10);    // tests: force slab overflows
                if (tuning_.use_rccl) { if (fqd_shard_unique_id(id) != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr)); cfg.unique_id = id; }
                if (fqd_shard_create(engines.data(), &cfg, &shard.g) != FQD_OK)
                    throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr));
                first_group = false;
            }
            // missing ranks of a short last round take part with no reads
            for (int r = 0; r < N; ++r)
                if (!round[size_t(r)]) for (int s = 0; s < S; ++s) { fqd_reads& d = seg[size_t(r) * size_t(S) + size_t(s)]; d = fqd_reads{}; d.uniform_len = s ? len1 : len0; d.uniform_stride = d.uniform_len; }
            shard_ok(fqd_shard_round(shard.g, seg.data(), n_of.data(), keep_of.data()));
            ++rounds_started;
            // the round before this one has its flags on the way now: hand it to the writers while this one runs
            if (rounds_started >= 2) deliver(previous, rounds_started - 2);
            previous = round;
            round.assign(size_t(N), nullptr);
        }
        if (rounds_started) {
            const int rc = fqd_shard_flush(shard.g);
            if (rc != FQD_OK && rc != FQD_ERR_BAD_BASE) shard_ok(rc);
            if (!previous.empty()) deliver(previous, rounds_started - 1);
        }
    } catch (...) {
        for (std::vector<Work*>* v : {&round, &previous})
            for (Work* w : *v) if (w) { for (int s = 0; s < S; ++s) if (w->blk[s]) w->blk[s]->release(); w->home->push(w); }
        if (shard.g) (void)fqd_shard_flush(shard.g);
        writers.stop(&stop_marker);
        throw;
    }
    { StageClock::Scope t("main: drain writers"); writers.stop(&stop_marker); }
    writers.rethrow();
    for (int s = 0; s < S; ++s) sink[s]->close();
    StageClock::report();
    for (auto& k : rank) { fqd_stats st{}; fqd_get_stats(k->eng->e, &st); total_dups += st.duplicates; }
    if (bad_base) throw_unknown_base(bad_byte);
    for (int s = 0; s < S; ++s) {
        if (side[s].available() == 0 && side[s].failed && side[s].held_back) {
            bool other_has = true;
            if (S == 2) other_has = side[1 - s].has_record_here();
            if (other_has) { std::cerr << side[s].failure.diag; throw std::runtime_error(side[s].failure.what); }
        }
    }
    summary_.total = next_index; summary_.duplicates = total_dups; summary_.unmatched = 0;
    if (verbose_) {
        if (S == 1) std::cout << summary_.total << " reads processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        else        std::cout << summary_.total << " read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
    }
}

// ---------------------------------------------------------------------------
// `--unordered` over several GPUs (FQD_DEVICES; SURVEY §8e: "a second exchange keyed by ID-tag precedes it (join), then
// the key exchange").  The reference sorts both files by ID tag and merge-joins them (hpp:150-192,257-347): one global
// tag order.  Here that order is cut into one RANGE per GPU:
//   1. the blocks of both files are dealt to the GPUs as they are read (text to HBM, records indexed, tags found);
//   2. splitters are picked from a sample of tags; every record — its whole text — moves to the GPU that owns its
//      tag's range (fqd_classify_tags, fqd_range_keep + fqd_output_plan + fqd_copy_spans, peer copies);
//   3. every GPU cuts what it received into records again and joins ITS range with the single-GPU join (fqd_join_tags):
//      the ranges' pair lists, one after the other, are the global pair list in tag order;
//   4. the reference's end-of-file rule looks at the global order through the ranges (reference_tail_rule);
//   5. the pairs are deduplicated through the shard group (csrc/fqd_shard.hip) in ONE round: rank = range, position =
//      tag order within it, so the group's global order (rank, position) IS the tag order and the smallest tag wins
//      (hpp:281-310); keys are padded (mates of any lengths);
//   6. the survivors are written range after range, each by the single-GPU writer (device codecs included).
void HashDupRemover::run_unordered_multi(const std::string* in, const std::string* out)
{
    const std::vector<int>& devs = tuning_.devices;
    const int N = static_cast<int>(devs.size());
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));
    struct Rank {
        int device = 0; hipStream_t stream = nullptr; std::unique_ptr<EngineHandle> eng;
        FileOnDevice part[2];                                  // what was dealt to this GPU of each file
        FileOnDevice range[2];                                 // this GPU's range of each file
        Device<uint32_t> cls[2];                               // range of every dealt record
        JoinedPairs jp; uint64_t n_pairs = 0, n_proc = 0;
        Device<uint64_t> d_off[2]; Device<uint32_t> d_len[2];
        bool joined = false;
        ~Rank() { (void)hipSetDevice(device); eng.reset(); if (stream) (void)hipStreamDestroy(stream); }
    };
    std::vector<std::unique_ptr<Rank>> rank;
    for (int r = 0; r < N; ++r) {
        rank.emplace_back(new Rank());
        Rank& k = *rank.back();
        k.device = devs[size_t(r)];
        HIP_OK(hipSetDevice(k.device));
        HIP_OK(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
        k.eng = std::make_unique<EngineHandle>(2, k.device, k.stream);
    }
    auto eng_ok = [&](Rank& k, int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(k.eng->e)); };

    // ---- 1. both files, block by block, dealt to the GPUs ---------------------------------------------------------
    {
        StageClock::Scope t("unordered/multi: read, scan, text to the GPUs");
        std::exception_ptr err[2];
        ParseFailure parse_failure[2];
        auto load = [&](int s) {
            try {
                std::vector<hipStream_t> up(size_t(N), nullptr);
                struct Guard { std::vector<hipStream_t>& v; std::vector<std::unique_ptr<Rank>>& rk; ~Guard() { for (size_t g = 0; g < v.size(); ++g) if (v[g]) { (void)hipSetDevice(rk[g]->device); (void)hipStreamDestroy(v[g]); } } } guard{up, rank};
                for (int g = 0; g < N; ++g) { HIP_OK(hipSetDevice(rank[size_t(g)]->device)); HIP_OK(hipStreamCreateWithFlags(&up[size_t(g)], hipStreamNonBlocking)); }
                Pinned<uint64_t> h_start, h_seq; Pinned<uint32_t> h_idl, h_sql, h_size;
                Side side;
                side.open_file(in[s], format_, true, block_bytes);
                HIP_OK(hipSetDevice(rank[0]->device));
                side.prime(3, rank[0]->device);
                uint64_t block_no = 0;
                while (side.available() > 0) {
                    const int g = int(block_no++ % uint64_t(N));
                    HIP_OK(hipSetDevice(rank[size_t(g)]->device));
                    FileOnDevice& f = rank[size_t(g)]->part[s];
                    hipStream_t st = up[size_t(g)];
                    PooledBlock* b = side.cur;
                    const size_t from = side.pos, nb = b->recs.size() - from;
                    const RecordRef* r = &b->recs[from];
                    const uint64_t text_lo = r[0].start, bytes = r[nb - 1].start + r[nb - 1].size - text_lo;
                    f.text.room_for(bytes + 64, st);
                    HIP_OK(hipMemcpyAsync(f.text.p + f.text.used, b->text.p + text_lo, bytes, hipMemcpyHostToDevice, st));
                    h_start.reserve(nb); h_seq.reserve(nb); h_idl.reserve(nb); h_sql.reserve(nb); h_size.reserve(nb);
                    for (size_t k = 0; k < nb; ++k) {
                        h_start.p[k] = f.text.used + (r[k].start - text_lo); h_seq.p[k] = h_start.p[k] + r[k].id_len;
                        h_idl.p[k] = r[k].id_len; h_sql.p[k] = r[k].seq_len; h_size.p[k] = r[k].size;
                    }
                    f.start.room_for(nb, st); f.seq_off.room_for(nb, st); f.id_len.room_for(nb, st); f.seq_len.room_for(nb, st); f.size.room_for(nb, st);
                    HIP_OK(hipMemcpyAsync(f.start.p + f.n, h_start.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipMemcpyAsync(f.seq_off.p + f.n, h_seq.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipMemcpyAsync(f.id_len.p + f.n, h_idl.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipMemcpyAsync(f.seq_len.p + f.n, h_sql.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipMemcpyAsync(f.size.p + f.n, h_size.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipStreamSynchronize(st));            // the block and the staging arrays are reused
                    f.text.used += bytes;
                    f.start.used = f.seq_off.used = f.id_len.used = f.seq_len.used = f.size.used = f.n + nb;
                    f.n += nb;
                    side.pos += nb;
                }
                if (side.failed) parse_failure[s] = side.failure;
            } catch (...) { err[s] = std::current_exception(); }
        };
        std::thread second(load, 1);
        load(0);
        second.join();
        for (int s = 0; s < 2; ++s) {                            // everything about file 1 before anything about file 2 (hpp:161-173)
            if (err[s]) std::rethrow_exception(err[s]);
            if (parse_failure[s].set) { std::cerr << parse_failure[s].diag; throw std::runtime_error(parse_failure[s].what); }
        }
    }
    uint64_t n_file[2] = {0, 0};
    for (auto& k : rank) for (int s = 0; s < 2; ++s) n_file[s] += k->part[s].n;
    if (n_file[0] >= 0x80000000ull || n_file[1] >= 0x80000000ull) throw std::runtime_error("--unordered: more than 2^31-1 records in one file");

    // ---- 2. tags, splitters from a sample, every record to the GPU of its range ------------------------------------
    constexpr uint32_t kSampleStride = 256;                      // bytes of a tag a splitter keeps
    std::vector<std::string> splitters;
    {
        StageClock::Scope t("unordered/multi: tags, splitters, records to their ranges");
        std::vector<std::string> sample;
        for (auto& kp : rank) {
            Rank& k = *kp;
            HIP_OK(hipSetDevice(k.device));
            for (int s = 0; s < 2; ++s) {
                FileOnDevice& f = k.part[s];
                if (!f.n) continue;
                f.tag_off.reserve(f.n); f.tag_len.reserve(f.n);
                eng_ok(k, fqd_extract_tags(k.eng->e, reinterpret_cast<const uint8_t*>(f.text.p), f.start.p, f.id_len.p, f.n, f.tag_off.p, f.tag_len.p));
                const uint32_t want = uint32_t(std::min<uint64_t>(f.n, 4096));
                Device<uint8_t> d_bytes; Device<uint32_t> d_len;
                d_bytes.reserve(size_t(want) * kSampleStride); d_len.reserve(want);
                const fqd_tags tg{reinterpret_cast<const uint8_t*>(f.text.p), f.tag_off.p, f.tag_len.p, f.n};
                eng_ok(k, fqd_sample_tags(k.eng->e, &tg, want, kSampleStride, d_bytes.p, d_len.p));
                std::vector<uint8_t> hb(size_t(want) * kSampleStride); std::vector<uint32_t> hl(want);
                HIP_OK(hipMemcpyAsync(hb.data(), d_bytes.p, hb.size(), hipMemcpyDeviceToHost, k.stream));
                HIP_OK(hipMemcpyAsync(hl.data(), d_len.p, hl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, k.stream));
                HIP_OK(hipStreamSynchronize(k.stream));
                for (uint32_t i = 0; i < want; ++i) sample.emplace_back(reinterpret_cast<const char*>(hb.data()) + size_t(i) * kSampleStride, hl[i]);
            }
        }
        // FastqViewWithId::cmp (fastqview.cpp:168-178) is what std::string's ordering is on unsigned bytes: bytes over the
        // shorter length, the shorter first on a tie
        auto tag_lt = [](const std::string& a, const std::string& b) {
            const size_t m = std::min(a.size(), b.size());
            const int c = std::memcmp(a.data(), b.data(), m);
            return c ? c < 0 : a.size() < b.size();
        };
        std::sort(sample.begin(), sample.end(), tag_lt);
        for (int j = 1; j < N && !sample.empty(); ++j) splitters.push_back(sample[sample.size() * size_t(j) / size_t(N)]);
        splitters.erase(std::unique(splitters.begin(), splitters.end()), splitters.end());
        const uint32_t n_split = uint32_t(splitters.size());
        std::vector<uint8_t> sb(size_t(std::max<uint32_t>(n_split, 1)) * kSampleStride, 0); std::vector<uint32_t> sl(std::max<uint32_t>(n_split, 1), 0);
        for (uint32_t j = 0; j < n_split; ++j) { std::memcpy(sb.data() + size_t(j) * kSampleStride, splitters[j].data(), splitters[j].size()); sl[j] = uint32_t(splitters[j].size()); }

        // how much of every (source GPU, file) goes to every range
        std::vector<uint64_t> bytes_to(size_t(N) * N * 2, 0), recs_to(size_t(N) * N * 2, 0);      // [(g * N + j) * 2 + s]
        struct Plan { Device<uint8_t> keep; Device<uint64_t> src_off, dst_off; Device<uint32_t> len; };
        for (int g = 0; g < N; ++g) {
            Rank& k = *rank[size_t(g)];
            HIP_OK(hipSetDevice(k.device));
            Device<uint8_t> d_sb; Device<uint32_t> d_sl;
            d_sb.reserve(sb.size()); d_sl.reserve(sl.size());
            HIP_OK(hipMemcpyAsync(d_sb.p, sb.data(), sb.size(), hipMemcpyHostToDevice, k.stream));
            HIP_OK(hipMemcpyAsync(d_sl.p, sl.data(), sl.size() * sizeof(uint32_t), hipMemcpyHostToDevice, k.stream));
            for (int s = 0; s < 2; ++s) {
                FileOnDevice& f = k.part[s];
                if (!f.n) continue;
                k.cls[s].reserve(f.n);
                const fqd_tags tg{reinterpret_cast<const uint8_t*>(f.text.p), f.tag_off.p, f.tag_len.p, f.n};
                eng_ok(k, fqd_classify_tags(k.eng->e, &tg, d_sb.p, kSampleStride, d_sl.p, n_split, k.cls[s].p));
            }
            HIP_OK(hipStreamSynchronize(k.stream));             // d_sb / d_sl go out of scope
        }
        // one range at a time: the pieces are cut out on their source GPU and copied to the tail of the range's text
        for (int j = 0; j < N; ++j) {
            Rank& own = *rank[size_t(j)];
            for (int s = 0; s < 2; ++s) {
                std::vector<std::unique_ptr<Device<char>>> piece(static_cast<size_t>(N));
                std::vector<uint64_t> piece_bytes(size_t(N), 0);
                uint64_t total = 0;
                for (int g = 0; g < N; ++g) {
                    Rank& k = *rank[size_t(g)];
                    FileOnDevice& f = k.part[s];
                    if (!f.n) continue;
                    HIP_OK(hipSetDevice(k.device));
                    Plan p; p.keep.reserve(f.n); p.src_off.reserve(f.n); p.dst_off.reserve(f.n + 1); p.len.reserve(f.n);
                    uint64_t cnt = 0, bytes = 0;
                    eng_ok(k, fqd_range_keep(k.eng->e, k.cls[s].p, f.n, uint32_t(j), p.keep.p, &cnt));
                    if (!cnt) continue;
                    eng_ok(k, fqd_output_plan(k.eng->e, p.keep.p, nullptr, f.n, f.start.p, f.size.p, p.src_off.p, p.len.p, p.dst_off.p, &bytes));
                    piece[size_t(g)] = std::make_unique<Device<char>>();
                    piece[size_t(g)]->reserve(bytes + 64);
                    eng_ok(k, fqd_copy_spans(k.eng->e, reinterpret_cast<const uint8_t*>(f.text.p), p.src_off.p, p.len.p, f.n,
                                             reinterpret_cast<uint8_t*>(piece[size_t(g)]->p), p.dst_off.p));
                    HIP_OK(hipStreamSynchronize(k.stream));     // the plan arrays go out of scope; the piece is complete
                    piece_bytes[size_t(g)] = bytes; total += bytes;
                    recs_to[(size_t(g) * N + size_t(j)) * 2 + size_t(s)] = cnt; bytes_to[(size_t(g) * N + size_t(j)) * 2 + size_t(s)] = bytes;
                }
                HIP_OK(hipSetDevice(own.device));
                FileOnDevice& dst = own.range[s];
                dst.text.room_for(total + 64, own.stream);
                uint64_t at = 0;
                for (int g = 0; g < N; ++g) {
                    if (!piece_bytes[size_t(g)]) continue;
                    const Rank& k = *rank[size_t(g)];
                    if (k.device == own.device) HIP_OK(hipMemcpyAsync(dst.text.p + at, piece[size_t(g)]->p, piece_bytes[size_t(g)], hipMemcpyDeviceToDevice, own.stream));
                    else                        HIP_OK(hipMemcpyPeerAsync(dst.text.p + at, own.device, piece[size_t(g)]->p, k.device, piece_bytes[size_t(g)], own.stream));
                    at += piece_bytes[size_t(g)];
                }
                HIP_OK(hipStreamSynchronize(own.stream));
                dst.text.used = total;
                for (int g = 0; g < N; ++g) if (piece[size_t(g)]) { HIP_OK(hipSetDevice(rank[size_t(g)]->device)); piece[size_t(g)].reset(); }
            }
        }
        // what was dealt has moved on
        for (auto& kp : rank) { HIP_OK(hipSetDevice(kp->device)); for (int s = 0; s < 2; ++s) { kp->part[s].release(); kp->cls[s].release(); } }
    }

    // ---- 3. every GPU: records of its range, tags, join ----------------------------------------------------------------
    std::vector<uint64_t> base_a(size_t(N) + 1, 0), base_b(size_t(N) + 1, 0), base_p(size_t(N) + 1, 0);
    {
        StageClock::Scope t("unordered/multi: record scan + tag join per range");
        for (int j = 0; j < N; ++j) {
            Rank& k = *rank[size_t(j)];
            HIP_OK(hipSetDevice(k.device));
            for (int s = 0; s < 2; ++s) {
                FileOnDevice& f = k.range[s];
                if (f.text.used == 0) { f.n = 0; continue; }
                const uint64_t bytes = f.text.used;
                if (!records_on_device(k.eng->e, k.stream, format_, bytes, f)) throw std::runtime_error("--unordered: internal: a range's text is not whole records");
                f.tag_off.reserve(f.n); f.tag_len.reserve(f.n);
                eng_ok(k, fqd_extract_tags(k.eng->e, reinterpret_cast<const uint8_t*>(f.text.p), f.start.p, f.id_len.p, f.n, f.tag_off.p, f.tag_len.p));
            }
            const uint64_t na = k.range[0].n, nb = k.range[1].n;
            base_a[size_t(j) + 1] = base_a[size_t(j)] + na; base_b[size_t(j) + 1] = base_b[size_t(j)] + nb;
            if (na && nb) {
                const uint64_t max_pairs = std::min(na, nb);
                for (int s = 0; s < 2; ++s) { k.jp.perm[s].reserve(k.range[s].n); k.jp.match[s].reserve(k.range[s].n); k.jp.pair[s].reserve(max_pairs); }
                const fqd_tags ta{reinterpret_cast<const uint8_t*>(k.range[0].text.p), k.range[0].tag_off.p, k.range[0].tag_len.p, na};
                const fqd_tags tb{reinterpret_cast<const uint8_t*>(k.range[1].text.p), k.range[1].tag_off.p, k.range[1].tag_len.p, nb};
                const fqd_join jo{k.jp.perm[0].p, k.jp.perm[1].p, k.jp.match[0].p, k.jp.match[1].p, k.jp.pair[0].p, k.jp.pair[1].p, &k.n_pairs};
                eng_ok(k, fqd_join_tags(k.eng->e, &ta, &tb, &jo));
                k.joined = true;
            }
            base_p[size_t(j) + 1] = base_p[size_t(j)] + k.n_pairs;
        }
    }
    if (base_a[size_t(N)] == 0 || base_b[size_t(N)] == 0) throw std::runtime_error("Not enough memory to read a single object!");   // (an empty input never gets here: the reader throws this)

    // ---- 4. the end-of-file rule over the global order ---------------------------------------------------------------------
    TailOutcome outcome{0, false, 0};
    {
        auto range_of = [&](const std::vector<uint64_t>& base, uint64_t pos) { int j = 0; while (j + 1 < N && base[size_t(j) + 1] <= pos) ++j; return j; };
        JoinLookup look;
        look.n = base_a[size_t(N)]; look.m = base_b[size_t(N)]; look.n_pairs = base_p[size_t(N)];
        look.match_a = [&](uint64_t i) -> uint32_t {
            const int j = range_of(base_a, i); Rank& k = *rank[size_t(j)];
            if (!k.joined) return kNoPartner;
            HIP_OK(hipSetDevice(k.device));
            const uint32_t v = peek_u32(k.jp.match[0].p, i - base_a[size_t(j)], k.stream);
            return v == kNoPartner ? kNoPartner : uint32_t(v + base_b[size_t(j)]);
        };
        look.match_b = [&](uint64_t i) -> uint32_t {
            const int j = range_of(base_b, i); Rank& k = *rank[size_t(j)];
            if (!k.joined) return kNoPartner;
            HIP_OK(hipSetDevice(k.device));
            const uint32_t v = peek_u32(k.jp.match[1].p, i - base_b[size_t(j)], k.stream);
            return v == kNoPartner ? kNoPartner : uint32_t(v + base_a[size_t(j)]);
        };
        // tags of the other file that are <= the tag at a sorted position: every lower range whole, and a count inside this one
        auto count_le = [&](int of, const std::vector<uint64_t>& base_other, const std::vector<uint64_t>& base_of, uint64_t pos_other) -> uint64_t {
            const int other = 1 - of;
            const int j = range_of(base_other, pos_other); Rank& k = *rank[size_t(j)];
            if (!k.range[of].n) return base_of[size_t(j)];
            HIP_OK(hipSetDevice(k.device));
            uint64_t c = 0;
            const fqd_tags tg[2] = {{reinterpret_cast<const uint8_t*>(k.range[0].text.p), k.range[0].tag_off.p, k.range[0].tag_len.p, k.range[0].n},
                                    {reinterpret_cast<const uint8_t*>(k.range[1].text.p), k.range[1].tag_off.p, k.range[1].tag_len.p, k.range[1].n}};
            const uint64_t local = pos_other - base_other[size_t(j)];
            eng_ok(k, fqd_count_tags_le(k.eng->e, &tg[of], &tg[other], peek_u32(k.jp.perm[other].p, local, k.stream), &c));
            return base_of[size_t(j)] + c;
        };
        look.count_b_le_a = [&](uint64_t i) { return count_le(1, base_a, base_b, i); };
        look.count_a_le_b = [&](uint64_t i) { return count_le(0, base_b, base_a, i); };
        outcome = tuning_.reference_tail_rule ? reference_tail_rule(look) : full_join_outcome(look);
    }
    for (auto& kp : rank) kp->n_proc = kp->n_pairs;
    if (outcome.drop_last) for (int j = N - 1; j >= 0; --j) if (rank[size_t(j)]->n_pairs) { rank[size_t(j)]->n_proc -= 1; break; }    // the last pair in tag order

    // outputs are opened after the sort phase (hpp:265-266)
    OutputFile sink0(out[0]), sink1(out[1]);
    OutputFile* sinks[2] = {&sink0, &sink1};

    // ---- 5. pair dedup through the shard group: one round, rank = range, position = tag order ---------------------------------
    bool bad = false; uint8_t bad_byte = 0; int bad_rank = N; uint64_t bad_at = 0;
    {
        StageClock::Scope t("unordered/multi: pair dedup over the GPUs");
        uint32_t max_len[2] = {1, 1};
        uint64_t most = 1;
        for (auto& kp : rank) {
            Rank& k = *kp;
            HIP_OK(hipSetDevice(k.device));
            most = std::max(most, k.n_proc);
            for (int s = 0; s < 2; ++s) {
                uint32_t m = 0;
                eng_ok(k, fqd_max_u32(k.eng->e, k.range[s].seq_len.p, k.range[s].n, &m));
                max_len[s] = std::max(max_len[s], m);
                if (!k.n_proc) continue;
                k.d_off[s].reserve(k.n_proc); k.d_len[s].reserve(k.n_proc);
                eng_ok(k, fqd_gather_seqs(k.eng->e, k.jp.pair[s].p, k.n_proc, k.range[s].seq_off.p, k.range[s].seq_len.p, k.d_off[s].p, k.d_len[s].p));
            }
            k.jp.keep.reserve(std::max<uint64_t>(k.n_proc, 1));
        }
        std::vector<fqd_engine*> engines;
        for (auto& kp : rank) engines.push_back(kp->eng->e);
        uint8_t id[FQD_SHARD_ID_BYTES] = {};
        fqd_shard_config cfg{};
        cfg.world = N; cfg.n_local = N; cfg.first_rank = 0;
        cfg.transport = tuning_.use_rccl ? FQD_SHARD_RCCL : FQD_SHARD_COPY;
        cfg.round_reads = most; cfg.len0 = max_len[0]; cfg.len1 = max_len[1]; cfg.flags = FQD_SHARD_PADDED;
        if (const char* v = std::getenv("FQD_SHARD_SLAB")) cfg.slab_records = std::strtoull(v, nullptr, 10);
        if (tuning_.use_rccl) { if (fqd_shard_unique_id(id) != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr)); cfg.unique_id = id; }
        struct ShardGuard { fqd_shard* g = nullptr; ~ShardGuard() { if (g) fqd_shard_destroy(g); } } shard;
        if (fqd_shard_create(engines.data(), &cfg, &shard.g) != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr));
        std::vector<fqd_reads> seg(size_t(N) * 2);
        std::vector<uint64_t> n_of(size_t(N), 0);
        std::vector<uint8_t*> keep_of(size_t(N), nullptr);
        for (int j = 0; j < N; ++j) {
            Rank& k = *rank[size_t(j)];
            for (int s = 0; s < 2; ++s) {
                fqd_reads& d = seg[size_t(j) * 2 + size_t(s)];
                d = fqd_reads{};
                d.bases = reinterpret_cast<const uint8_t*>(k.range[s].text.p); d.offsets = k.d_off[s].p; d.lengths = k.d_len[s].p;
                if (!k.n_proc) { d.offsets = nullptr; d.lengths = nullptr; d.uniform_len = max_len[s]; d.uniform_stride = max_len[s]; }
            }
            n_of[size_t(j)] = k.n_proc; keep_of[size_t(j)] = k.jp.keep.p;
        }
        auto shard_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(shard.g)); };
        shard_ok(fqd_shard_round(shard.g, seg.data(), n_of.data(), keep_of.data()));
        int rc = fqd_shard_flush(shard.g);
        if (rc != FQD_OK && rc != FQD_ERR_BAD_BASE) shard_ok(rc);
        rc = fqd_shard_wait(shard.g, 0);
        if (rc == FQD_ERR_BAD_BASE) {
            int32_t lr = 0; uint64_t rec = 0; uint32_t sg = 0, pos = 0;
            if (fqd_shard_bad_base(shard.g, 0, &lr, &rec, &sg, &pos, &bad_byte) == FQD_OK) { bad = true; bad_rank = lr; bad_at = rec; }
        } else shard_ok(rc);
    }

    // ---- 6. survivors, range after range -------------------------------------------------------------------------------------------
    uint64_t total = 0, dups = 0;
    {
        StageClock::Scope t("unordered/multi: survivors out of HBM");
        for (int j = 0; j < N; ++j) {
            Rank& k = *rank[size_t(j)];
            total += k.n_proc;
            uint64_t upto = k.n_proc;
            if (bad && j == bad_rank) upto = std::min(upto, bad_at);   // the output is cut at the pair that held the bad byte
            if (bad && j > bad_rank) upto = 0;
            if (!upto) continue;
            HIP_OK(hipSetDevice(k.device));
            std::vector<uint8_t> keep(upto);
            HIP_OK(hipMemcpyAsync(keep.data(), k.jp.keep.p, upto, hipMemcpyDeviceToHost, k.stream));
            HIP_OK(hipStreamSynchronize(k.stream));
            uint64_t d = 0;
            for (uint64_t q = 0; q < upto; ++q) d += keep[q] == 0;
            dups += d;
            FileOnDevice* files[2] = {&k.range[0], &k.range[1]};
            const uint32_t* idx[2] = {k.jp.pair[0].p, k.jp.pair[1].p};
            write_survivors(k.eng->e, k.stream, 2, files, idx, k.jp.keep.p, upto, d, sinks, format_, memlimit_, false);
        }
        sink0.close(); sink1.close();
    }
    StageClock::report();
    if (bad) throw_unknown_base(bad_byte);
    summary_.total = total; summary_.duplicates = dups; summary_.unmatched = outcome.unmatched;
    if (verbose_) {
        std::cout << summary_.total << " valid read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        std::cout << summary_.unmatched << " Non-matching entries from both files were skipped.\n";
    }
}

} // namespace fqdhost
