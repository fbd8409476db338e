// run_unordered.cpp — `--unordered` on one GPU (reference hash_dup_remover.hpp:150-192,257-347): the device stage all ways
// of running share (join_and_dedup), the dispatch, the in-memory cross-check run and the two-pass bounded-memory run.
#include "run_common.hpp"

namespace fqdhost {
using namespace detail;

namespace detail {

// Copies entry k of a device array of uint32.
uint32_t peek_u32(const uint32_t* d, uint64_t k, hipStream_t s)
{
    uint32_t v = 0;
    HIP_OK(hipMemcpyAsync(&v, d + k, sizeof v, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return v;
}

void join_and_dedup(fqd_engine* e, hipStream_t stream, const DeviceSide (&side)[2], bool tail_rule, JoinedPairs& jp)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    if (side[0].n >= 0x80000000ull || side[1].n >= 0x80000000ull)
        throw std::runtime_error("--unordered: more than 2^31-1 records in one file");
    const uint64_t max_pairs = std::min(side[0].n, side[1].n);
    uint64_t n_pairs = 0;
    TailOutcome outcome{0, false, 0};
    const fqd_tags tags[2] = {{side[0].tag_bytes, side[0].tag_off, side[0].tag_len, side[0].n},
                              {side[1].tag_bytes, side[1].tag_off, side[1].tag_len, side[1].n}};
    {
        StageClock::Scope t("unordered: tag join on the GPU");
        for (int s = 0; s < 2; ++s) { jp.perm[s].reserve(side[s].n); jp.match[s].reserve(side[s].n); jp.pair[s].reserve(max_pairs); }
        const fqd_join jo{jp.perm[0].p, jp.perm[1].p, jp.match[0].p, jp.match[1].p, jp.pair[0].p, jp.pair[1].p, &n_pairs};
        engine_ok(fqd_join_tags(e, &tags[0], &tags[1], &jo));
        JoinLookup look;
        look.n = side[0].n; look.m = side[1].n; look.n_pairs = n_pairs;
        look.match_a = [&](uint64_t k) { return peek_u32(jp.match[0].p, k, stream); };
        look.match_b = [&](uint64_t k) { return peek_u32(jp.match[1].p, k, stream); };
        // "how many tags of the other file are <= the tag at this sorted position": a count over the other file's tags
        auto count_le = [&](int of, uint64_t pos_other) {
            const int other = 1 - of;
            uint64_t c = 0;
            engine_ok(fqd_count_tags_le(e, &tags[of], &tags[other], peek_u32(jp.perm[other].p, pos_other, stream), &c));
            return c;
        };
        look.count_b_le_a = [&](uint64_t i) { return count_le(1, i); };
        look.count_a_le_b = [&](uint64_t j) { return count_le(0, j); };
        outcome = tail_rule ? reference_tail_rule(look) : full_join_outcome(look);
    }
    jp.n_proc = outcome.pairs; jp.unmatched = outcome.unmatched; jp.written_below = outcome.pairs;
    // pair-dedup in tag order: the pairs' sequences are read where they lie through offset/length
    // arrays gathered on the device; batches are queued back to back, the host waits once
    StageClock::Scope t("unordered: pair dedup on the GPU");
    const uint64_t n_proc = jp.n_proc;
    Device<uint64_t>* d_off = jp.seq_off; Device<uint32_t>* d_len = jp.seq_len;
    for (int s = 0; s < 2; ++s) {
        d_off[s].reserve(n_proc); d_len[s].reserve(n_proc);
        engine_ok(fqd_gather_seqs(e, jp.pair[s].p, n_proc, side[s].seq_off, side[s].seq_len, d_off[s].p, d_len[s].p));
    }
    jp.keep.reserve(n_proc);
    const size_t kBatch = 16u << 20;
    int rc = FQD_OK;
    for (size_t a = 0; a < n_proc && rc == FQD_OK; a += kBatch) {
        const size_t n = std::min<size_t>(kBatch, n_proc - a);
        fqd_reads seg[2] = {};
        for (int s = 0; s < 2; ++s) { seg[s].bases = side[s].seq_bytes; seg[s].offsets = d_off[s].p + a; seg[s].lengths = d_len[s].p + a; }
        // the last batch says so: the set is never looked at again, so its segments are not written back to HBM
        rc = (a + n < n_proc ? fqd_submit : fqd_submit_final)(e, seg, n, FQD_MEM_DEVICE, jp.keep.p + a);
    }
    if (rc == FQD_OK) rc = fqd_engine_sync(e);
    if (rc == FQD_ERR_BAD_BASE) {
        uint64_t rec; uint32_t sg2, pos;
        fqd_bad_base(e, &rec, &sg2, &pos, &jp.bad_byte);
        jp.bad = true; jp.written_below = std::min<uint64_t>(rec, n_proc);
    } else engine_ok(rc);
}

bool is_regular_file(const std::string& name, uint64_t& size)
{
    std::error_code ec;
    const auto st = std::filesystem::status(name, ec);
    if (ec || !std::filesystem::is_regular_file(st)) return false;
    size = std::filesystem::file_size(name, ec);
    return !ec;
}

} // namespace detail

// ---------------------------------------------------------------------------
// --unordered (hash_dup_remover.hpp:150-192,257-347): join the two files on the ID tag,
// dedup the joined pairs in tag order, write survivors in tag order.
//
// The reference bounds its memory here with ExternalSorter(memlimit) (hpp:165,171;
// external_sort.hpp:95): sorted chunk files on disk, merged.  This build:
//   * streams both files ONCE through a few pinned blocks (each at most limit/16 bytes) into HBM, where the
//     whole text of both files stays (run_unordered_resident; configs[4]: 2 x 32 GB of 288): the device joins,
//     dedups and then assembles the outputs window by window in output order; the host only reads, and writes
//     what comes back;
//   * what not even HBM can hold is streamed TWICE (run_unordered_streaming): the first pass leaves every
//     record's tag and sequence in HBM (about 190 bytes per 150-bp record), the device decides everything
//     (pairs, survivors, where every surviving record starts in the output), the second pass puts the records
//     there through window files in the temporary directory.
//   * FQD_UNORDERED_MODE=memory: round 1's way — both files also held in pinned host memory, survivors written
//     from there (run_unordered_in_memory); kept as a cross-check of the other two.
// Host memory stays within the limit whatever the input size in the first two.
void HashDupRemover::run_unordered(const std::string* in, const std::string* out)
{
    uint64_t sz[2] = {0, 0};
    const bool regular = is_regular_file(in[0], sz[0]) && is_regular_file(in[1], sz[1]);
    std::string forced;
    if (const char* m = std::getenv("FQD_UNORDERED_MODE")) forced = m;
    if (!tuning_.devices.empty()) { run_unordered_multi(in, out); return; }
    if (forced == "memory") { run_unordered_in_memory(in, out); return; }
    if (forced == "twopass" && regular) { run_unordered_streaming(in, out); return; }
    // The text of both files goes to HBM block by block and stays there (one pass, nothing kept on the
    // host: within any --mem-limit, pipes included).  Only when 288 GB cannot hold it are tags and
    // sequences alone kept and the inputs read a second time.
    try { run_unordered_resident(in, out); }
    catch (const DeviceOutOfMemory&) {
        if (!regular) throw;
        run_unordered_streaming(in, out);
    }
}

void HashDupRemover::run_unordered_in_memory(const std::string* in, const std::string* out)
{
    HIP_OK(hipSetDevice(tuning_.device));
    // 1. load + index both files (the reference's ExternalSorter reads them fully too, hpp:161-173)
    LoadedFile file[2];
    {
        // both files are read (and, for .gz, inflated) at the same time; problems are still
        // reported in the reference's order: everything about file 1 before anything about file 2
        std::exception_ptr err[2];
        auto load = [&](int s) {
            (void)hipSetDevice(tuning_.device);                  // pinned chunks belong to this device's context
            try { load_whole_file(in[s], format_, tuning_.block_bytes, file[s]); }
            catch (...) { err[s] = std::current_exception(); }
        };
        StageClock::Scope t("unordered: load + index both files");
        std::thread second(load, 1);
        load(0);
        second.join();
        for (int s = 0; s < 2; ++s) {
            if (err[s]) std::rethrow_exception(err[s]);
            if (file[s].failure.set) { std::cerr << file[s].failure.diag; throw std::runtime_error(file[s].failure.what); }
        }
    }

    // 2. outputs are opened after the sort phase (hpp:265-266)
    OutputFile sink0(out[0]), sink1(out[1]);

    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    EngineHandle eng(2, tuning_.device, stream);
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.e)); };

    const uint64_t n_rec[2] = {file[0].recs.size(), file[1].recs.size()};

    // 3. text + per-record index to HBM; the tags are found on the device in the uploaded text
    Device<char> d_text[2]; Device<uint64_t> d_seq_off[2]; Device<uint32_t> d_id_len[2], d_seq_len[2];
    Device<uint64_t> d_id_start[2], d_tag_off[2]; Device<uint32_t> d_tag_len[2];
    {
        StageClock::Scope t("unordered: text + index to the GPU");
        size_t text_bytes[2] = {0, 0};
        for (int s = 0; s < 2; ++s) for (size_t u : file[s].chunk_used) text_bytes[s] += u;
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        const size_t n_all = n_rec[0] + n_rec[1];
        const size_t need = text_bytes[0] + text_bytes[1] + n_all * (40 + 40) + std::min(n_rec[0], n_rec[1]) * 230 + (size_t(2) << 30);
        if (need > free_b)
            throw std::runtime_error("--unordered: the two inputs (" + std::to_string((text_bytes[0] + text_bytes[1]) >> 20) +
                                     " MiB of text) do not fit in GPU memory beside the join and the set");
        Pinned<uint64_t> h_off, h_ids; Pinned<uint32_t> h_idl, h_sql;
        for (int s = 0; s < 2; ++s) {
            d_text[s].reserve(text_bytes[s] + 64);
            std::vector<uint64_t> chunk_base;
            uint64_t at = 0;
            for (size_t c = 0; c < file[s].chunks.size(); ++c) {
                chunk_base.push_back(at);
                HIP_OK(hipMemcpyAsync(d_text[s].p + at, file[s].chunks[c]->p, file[s].chunk_used[c], hipMemcpyHostToDevice, stream));
                at += file[s].chunk_used[c];
            }
            const size_t n = n_rec[s];
            h_off.reserve(n); h_ids.reserve(n); h_idl.reserve(n); h_sql.reserve(n);
            const unsigned parts = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(host_threads(), n >> 16)));
            run_parts(parts, [&](unsigned p) {
                for (size_t k = n / parts * p, e = p + 1 == parts ? n : n / parts * (p + 1); k < e; ++k) {
                    const FileRecord& r = file[s].recs[k];
                    h_ids.p[k] = chunk_base[r.chunk] + static_cast<uint64_t>(r.text - file[s].chunks[r.chunk]->p);
                    h_off.p[k] = h_ids.p[k] + r.id_len;
                    h_idl.p[k] = r.id_len; h_sql.p[k] = r.seq_len;
                }
            });
            d_seq_off[s].reserve(n); d_id_len[s].reserve(n); d_seq_len[s].reserve(n);
            d_id_start[s].reserve(n); d_tag_off[s].reserve(n); d_tag_len[s].reserve(n);
            HIP_OK(hipMemcpyAsync(d_seq_off[s].p, h_off.p, n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(d_id_len[s].p, h_idl.p, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(d_seq_len[s].p, h_sql.p, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(d_id_start[s].p, h_ids.p, n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            engine_ok(fqd_extract_tags(eng.e, reinterpret_cast<const uint8_t*>(d_text[s].p), d_id_start[s].p, d_id_len[s].p, n,
                                       d_tag_off[s].p, d_tag_len[s].p));
            HIP_OK(hipStreamSynchronize(stream));                // the pinned staging arrays are reused by file 2
        }
    }

    // 4. join + pair dedup on the GPU
    DeviceSide side[2];
    for (int s = 0; s < 2; ++s) {
        side[s].tag_bytes = side[s].seq_bytes = reinterpret_cast<const uint8_t*>(d_text[s].p);
        side[s].tag_off = d_tag_off[s].p; side[s].tag_len = d_tag_len[s].p;
        side[s].seq_off = d_seq_off[s].p; side[s].seq_len = d_seq_len[s].p; side[s].n = n_rec[s];
    }
    JoinedPairs jp;
    join_and_dedup(eng.e, stream, side, tuning_.reference_tail_rule, jp);
    const uint64_t n_proc = jp.n_proc;
    std::vector<uint8_t> keep(n_proc);
    std::vector<uint32_t> pair_idx[2];
    for (int s = 0; s < 2; ++s) {
        pair_idx[s].resize(n_proc);
        if (n_proc) HIP_OK(hipMemcpyAsync(pair_idx[s].data(), jp.pair[s].p, n_proc * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    if (n_proc) HIP_OK(hipMemcpyAsync(keep.data(), jp.keep.p, n_proc, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    // 5. survivors in tag order: one thread per output file, records handed over where they lie
    uint64_t dups = 0;
    {
        StageClock::Scope t("unordered: write survivors");
        const uint64_t upto = std::min<uint64_t>(n_proc, jp.written_below);
        for (uint64_t k = 0; k < upto; ++k) dups += keep[k] == 0;
        OutputFile* sinks[2] = {&sink0, &sink1};
        run_parts(2, [&](unsigned s) {
            std::vector<OutputFile::Piece> pieces;
            pieces.reserve(1u << 16);
            for (uint64_t k = 0; k < upto; ++k) {
                if (!keep[k]) continue;
                const FileRecord& r = file[s].recs[pair_idx[s][k]];
                pieces.push_back({r.text, r.size});
                if (pieces.size() == (1u << 16)) { sinks[s]->write_pieces(pieces.data(), pieces.size()); pieces.clear(); }
            }
            sinks[s]->write_pieces(pieces.data(), pieces.size());
            sinks[s]->close();
        });
    }
    StageClock::report();
    if (jp.bad) throw_unknown_base(jp.bad_byte);
    summary_.total = n_proc; summary_.duplicates = dups; summary_.unmatched = jp.unmatched;
    if (verbose_) {
        std::cout << summary_.total << " valid read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        std::cout << summary_.unmatched << " Non-matching entries from both files were skipped.\n";
    }
}

// The bounded-memory way (see run_unordered).
void HashDupRemover::run_unordered_streaming(const std::string* in, const std::string* out)
{
    HIP_OK(hipSetDevice(tuning_.device));
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    EngineHandle eng(2, tuning_.device, stream);
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.e)); };
    // pinned blocks in flight: three per file at most, sized so that they stay well inside the limit
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));

    struct FileOnDevice {
        GrowDevice<char> tags, seqs;
        GrowDevice<uint64_t> tag_off, seq_off; GrowDevice<uint32_t> tag_len, seq_len, size;
        uint64_t n = 0;
    } dev[2];

    // ---- pass 1: tags and sequences of every record into HBM -------------------------------------
    {
        StageClock::Scope t("unordered/stream: pass 1 (tags + sequences to HBM)");
        Device<char> d_block;
        Pinned<uint64_t> h_src_seq, h_src_tag, h_dst_seq, h_dst_tag; Pinned<uint32_t> h_seq_len, h_tag_len, h_size;
        Device<uint64_t> d_src_seq, d_src_tag;
        for (int s = 0; s < 2; ++s) {                          // file 1 completely before file 2 is touched, like the two sorts (hpp:161-173)
            FileOnDevice& f = dev[s];
            Side side;
            side.open_file(in[s], format_, true, block_bytes);
            side.prime(3, tuning_.device);
            while (side.available() > 0) {
                PooledBlock* b = side.cur;
                const size_t from = side.pos, nb = b->recs.size() - from;
                const RecordRef* r = &b->recs[from];
                const uint64_t text_lo = r[0].start, text_hi = r[nb - 1].start + r[nb - 1].size;
                d_block.reserve(text_hi - text_lo + 64);
                HIP_OK(hipMemcpyAsync(d_block.p, b->text.p + text_lo, text_hi - text_lo, hipMemcpyHostToDevice, stream));
                h_src_seq.reserve(nb); h_src_tag.reserve(nb); h_dst_seq.reserve(nb); h_dst_tag.reserve(nb);
                h_seq_len.reserve(nb); h_tag_len.reserve(nb); h_size.reserve(nb);
                uint64_t seq_at = f.seqs.used, tag_at = f.tags.used;
                for (size_t k = 0; k < nb; ++k) {
                    h_src_seq.p[k] = r[k].seq_start() - text_lo; h_seq_len.p[k] = r[k].seq_len; h_dst_seq.p[k] = seq_at; seq_at += r[k].seq_len;
                    h_src_tag.p[k] = r[k].start + r[k].tag_off - text_lo; h_tag_len.p[k] = r[k].tag_len; h_dst_tag.p[k] = tag_at; tag_at += r[k].tag_len;
                    h_size.p[k] = r[k].size;
                }
                f.seqs.room_for(seq_at - f.seqs.used + 16, stream); f.tags.room_for(tag_at - f.tags.used + 16, stream);
                f.seq_off.room_for(nb, stream); f.tag_off.room_for(nb, stream); f.seq_len.room_for(nb, stream); f.tag_len.room_for(nb, stream); f.size.room_for(nb, stream);
                d_src_seq.reserve(nb); d_src_tag.reserve(nb);
                HIP_OK(hipMemcpyAsync(d_src_seq.p, h_src_seq.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(d_src_tag.p, h_src_tag.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(f.seq_off.p + f.n, h_dst_seq.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(f.tag_off.p + f.n, h_dst_tag.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(f.seq_len.p + f.n, h_seq_len.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(f.tag_len.p + f.n, h_tag_len.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                HIP_OK(hipMemcpyAsync(f.size.p + f.n, h_size.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                engine_ok(fqd_copy_spans(eng.e, reinterpret_cast<const uint8_t*>(d_block.p), d_src_seq.p, f.seq_len.p + f.n, nb,
                                         reinterpret_cast<uint8_t*>(f.seqs.p), f.seq_off.p + f.n));
                engine_ok(fqd_copy_spans(eng.e, reinterpret_cast<const uint8_t*>(d_block.p), d_src_tag.p, f.tag_len.p + f.n, nb,
                                         reinterpret_cast<uint8_t*>(f.tags.p), f.tag_off.p + f.n));
                HIP_OK(hipStreamSynchronize(stream));            // the block and the staging arrays are reused
                f.seqs.used = seq_at; f.tags.used = tag_at;
                f.seq_off.used = f.tag_off.used = f.seq_len.used = f.tag_len.used = f.size.used = f.n + nb;
                f.n += nb;
                side.pos += nb;
            }
            if (side.failed) { std::cerr << side.failure.diag; throw std::runtime_error(side.failure.what); }
        }
    }

    // outputs are opened after the sort phase (hpp:265-266)
    OutputFile sink0(out[0]), sink1(out[1]);
    OutputFile* sinks[2] = {&sink0, &sink1};

    // ---- the device decides: pairs, survivors, where each survivor starts in its output file -------
    DeviceSide side[2];
    for (int s = 0; s < 2; ++s) {
        side[s].tag_bytes = reinterpret_cast<const uint8_t*>(dev[s].tags.p); side[s].tag_off = dev[s].tag_off.p; side[s].tag_len = dev[s].tag_len.p;
        side[s].seq_bytes = reinterpret_cast<const uint8_t*>(dev[s].seqs.p); side[s].seq_off = dev[s].seq_off.p; side[s].seq_len = dev[s].seq_len.p;
        side[s].n = dev[s].n;
    }
    JoinedPairs jp;
    join_and_dedup(eng.e, stream, side, tuning_.reference_tail_rule, jp);
    const uint64_t n_proc = jp.n_proc, upto = std::min<uint64_t>(n_proc, jp.written_below);
    uint64_t dups = 0;
    {
        std::vector<uint8_t> keep(upto);
        if (upto) HIP_OK(hipMemcpyAsync(keep.data(), jp.keep.p, upto, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (uint64_t k = 0; k < upto; ++k) dups += keep[k] == 0;
    }
    Device<uint64_t> d_dest[2];
    uint64_t out_bytes[2] = {0, 0};
    for (int s = 0; s < 2; ++s) {
        d_dest[s].reserve(dev[s].n);
        if (dev[s].n) HIP_OK(hipMemsetAsync(d_dest[s].p, 0xFF, dev[s].n * sizeof(uint64_t), stream));
        engine_ok(fqd_output_offsets(eng.e, jp.keep.p, jp.pair[s].p, upto, dev[s].size.p, d_dest[s].p, &out_bytes[s]));
    }

    // ---- pass 2: every surviving record to its place ------------------------------------------------
    {
        StageClock::Scope t("unordered/stream: pass 2 (records to their place in the outputs)");
        uint64_t window = std::max<uint64_t>(8u << 20, static_cast<uint64_t>(memlimit_ > 0 ? memlimit_ : (2ll << 30)) / 4);   // per file
        if (const char* v = std::getenv("FQD_STREAM_WINDOW_KB")) { const long kb = std::atol(v); if (kb > 0) window = static_cast<uint64_t>(kb) << 10; }   // tests: many windows on small inputs
        const std::string tmp = out_bytes[0] > window || out_bytes[1] > window ? std::string(tempdir_->name()) : std::string();
        run_parts(2, [&](unsigned s) {
            (void)hipSetDevice(tuning_.device);
            hipStream_t st2 = nullptr;
            HIP_OK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
            struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{st2};
            const uint64_t total = out_bytes[s];
            const size_t n_win = static_cast<size_t>((total + window - 1) / window);
            // windows of the output: window w holds the records that START in [w*window, (w+1)*window); the
            // records tile the output, so window w's bytes are [first start in w, first start in w+1)
            std::vector<char> direct;                           // the whole output fits one window: placed in memory
            std::vector<FILE*> spill(n_win > 1 ? n_win : 0, nullptr);
            std::vector<uint64_t> win_lo(n_win, ~0ull), win_hi(n_win, 0);
            if (n_win == 1) direct.resize(total);
            for (size_t w = 0; w < spill.size(); ++w) {
                const std::string name = tmp + "/out" + std::to_string(s) + "." + std::to_string(w) + ".tmp";
                spill[w] = std::fopen(name.c_str(), "wb+");
                if (!spill[w]) throw std::runtime_error("Cannot open temporary file " + name);
                std::setvbuf(spill[w], nullptr, _IOFBF, 1u << 20);
            }
            RecordStream rs(in[s], format_, false, block_bytes);
            Block b; Pinned<uint64_t> h_dest;
            uint64_t g0 = 0;
            while (rs.fill(b)) {
                const size_t nb = b.recs.size();
                if (nb) {
                    if (g0 + nb > dev[s].n) throw std::runtime_error("--unordered: " + in[s] + " changed between the two passes");
                    h_dest.reserve(nb);
                    HIP_OK(hipMemcpyAsync(h_dest.p, d_dest[s].p + g0, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, st2));
                    HIP_OK(hipStreamSynchronize(st2));
                    for (size_t k = 0; k < nb; ++k) {
                        const uint64_t at = h_dest.p[k];
                        if (at == ~0ull) continue;
                        const RecordRef& r = b.recs[k];
                        if (n_win == 1) { std::memcpy(direct.data() + at, b.text.p + r.start, r.size); continue; }
                        const size_t w = static_cast<size_t>(at / window);
                        win_lo[w] = std::min(win_lo[w], at); win_hi[w] = std::max<uint64_t>(win_hi[w], at + r.size);
                        const uint32_t size = r.size;
                        if (std::fwrite(&at, sizeof at, 1, spill[w]) != 1 || std::fwrite(&size, sizeof size, 1, spill[w]) != 1 ||
                            std::fwrite(b.text.p + r.start, 1, size, spill[w]) != size)
                            throw std::runtime_error("write failed: temporary file of " + out[s]);
                    }
                    g0 += nb;
                }
                if (b.last) break;
            }
            if (n_win == 1) sinks[s]->write(direct.data(), direct.size());
            std::vector<char> buf;
            for (size_t w = 0; w < spill.size(); ++w) {
                if (win_hi[w] > win_lo[w]) {
                    buf.assign(win_hi[w] - win_lo[w], 0);
                    std::rewind(spill[w]);
                    uint64_t at; uint32_t size;
                    while (std::fread(&at, sizeof at, 1, spill[w]) == 1) {
                        if (std::fread(&size, sizeof size, 1, spill[w]) != 1 || std::fread(buf.data() + (at - win_lo[w]), 1, size, spill[w]) != size)
                            throw std::runtime_error("read failed: temporary file of " + out[s]);
                    }
                    sinks[s]->write(buf.data(), buf.size());
                }
                std::fclose(spill[w]); spill[w] = nullptr;
            }
            sinks[s]->close();
        });
    }
    StageClock::report();
    if (jp.bad) throw_unknown_base(jp.bad_byte);
    summary_.total = n_proc; summary_.duplicates = dups; summary_.unmatched = jp.unmatched;
    if (verbose_) {
        std::cout << summary_.total << " valid read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        std::cout << summary_.unmatched << " Non-matching entries from both files were skipped.\n";
    }
}

} // namespace fqdhost
