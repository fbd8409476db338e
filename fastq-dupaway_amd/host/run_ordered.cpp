// run_ordered.cpp — the streaming ordered runs (single-end, paired files side by side) on one GPU, the temporary
// directory, and the dispatch of filterSE / filterPE to the ways of running (reference hash_dup_remover.hpp:97-255).
#include "run_common.hpp"

namespace fqdhost {
using namespace detail;

// ---------------------------------------------------------------------------
// TemporaryDirectory (file_utils.cpp:26-40,116-130), created on first use.
const char* TemporaryDirectory::name()
{
    if (name_.empty()) {
        static const char charset[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz";
        std::mt19937 rng(std::random_device{}());
        std::uniform_int_distribution<size_t> pick(0, sizeof(charset) - 2);
        for (int tries = 0; tries < 10; ++tries) {
            std::string cand(10, '0');                       // constants.hpp:9 DIRNAME_LEN
            for (char& c : cand) c = charset[pick(rng)];
            if (std::filesystem::create_directory(cand)) { name_ = cand; break; }
        }
        if (name_.empty()) throw std::runtime_error("Number of tries exhausted.");
    }
    return name_.c_str();
}

TemporaryDirectory::~TemporaryDirectory()
{
    if (!name_.empty()) { std::error_code ec; std::filesystem::remove_all(name_, ec); }
}

// ---------------------------------------------------------------------------
// Ordered runs: SE (hash_dup_remover.hpp:105-148) and PE (hash_dup_remover.hpp:194-255).
void HashDupRemover::run_ordered(int S, const std::string* in, const std::string* out)
{
    // Outputs are created before the inputs are opened (hpp:110,202-203), so an unreadable
    // input still leaves (empty) output files behind, as in the reference.
    std::unique_ptr<OutputFile> sink[2];
    for (int s = 0; s < S; ++s) sink[s] = std::make_unique<OutputFile>(out[s]);

    Side side[2];
    for (int s = 0; s < S; ++s) {
        side[s].open_file(in[s], format_, S == 2, tuning_.block_bytes);      // "Cannot open file" comes first
        HIP_OK(hipSetDevice(tuning_.device));
        side[s].prime(4, tuning_.device);
        // A malformed FIRST record fails at open, before anything is processed and before the
        // next file is touched (bufferedinput.hpp:38-42,81-84; hpp:211-212).
        if (side[s].available() == 0 && side[s].failed && !side[s].held_back) {
            std::cerr << side[s].failure.diag;
            throw std::runtime_error(side[s].failure.what);
        }
    }

    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    std::unique_ptr<EngineHandle> eng_holder;
    { StageClock::Scope t("main: engine create"); eng_holder = std::make_unique<EngineHandle>(S, tuning_.device, stream); }
    EngineHandle& eng = *eng_holder;

    constexpr int kWorks = 3;
    std::vector<std::unique_ptr<Work>> works;
    Channel<Work> free_works;
    for (int k = 0; k < kWorks; ++k) { works.emplace_back(new Work()); works.back()->S = S; free_works.push(works.back().get()); }

    SurvivorWriters writers(S, sink, &free_works);
    auto hand_to_writers = [&](Work* w) { writers.hand_over(w); };
    auto stop_writer = [&] { writers.stop(free_works.pop()); };

    uint64_t next_index = 0;
    bool bad_base = false; uint8_t bad_byte = 0; uint64_t bad_record = 0;
    Work* inflight = nullptr;
    constexpr size_t kMaxBatch = 8u << 20;                 // records per submit

    auto finish = [&](Work* w) {
        // waits for the batch; on an unknown base cuts the output at that record
        int rc;
        { StageClock::Scope t("main: wait for the GPU"); rc = fqd_engine_sync(eng.e); }
        if (rc == FQD_ERR_BAD_BASE) {
            uint32_t seg, pos;
            fqd_bad_base(eng.e, &bad_record, &seg, &pos, &bad_byte);
            bad_base = true;
            w->emit_below = bad_record;
        } else if (rc != FQD_OK) {
            for (int s = 0; s < S; ++s) w->blk[s]->release();
            free_works.push(w);
            throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.e));
        }
        hand_to_writers(w);
    };

    try {
        while (!bad_base) {
            size_t n = kMaxBatch;
            for (int s = 0; s < S; ++s) n = std::min(n, side[s].available());
            if (n == 0) break;
            Work* w;
            { StageClock::Scope t("main: wait for a free batch"); w = free_works.pop(); }
            StageClock::Scope prep("main: prepare + enqueue batch");
            w->stop = false; w->n = n; w->first_index = next_index; w->emit_below = ~0ull;
            fqd_reads seg[2] = {};
            w->keep.reserve(n); w->d_keep.reserve(n);
            for (int s = 0; s < S; ++s) {
                PooledBlock* b = side[s].cur;
                b->acquire();
                w->blk[s] = b; w->begin[s] = side[s].pos;
                const RecordRef* r = &b->recs[side[s].pos];
                const uint64_t text_lo = r[0].start;
                const uint64_t text_hi = r[n - 1].start + r[n - 1].size;
                // uniform batch: same sequence length everywhere and equally spaced sequences
                bool uniform = n > 1;
                const uint64_t stride = n > 1 ? r[1].seq_start() - r[0].seq_start() : 0;
                for (size_t k = 1; k < n && uniform; ++k)
                    uniform = r[k].seq_len == r[0].seq_len && r[k].seq_start() - r[k - 1].seq_start() == stride;
                uniform = uniform && stride <= 0xFFFFFFFFull;
                w->d_text[s].reserve(text_hi - text_lo + 32);
                HIP_OK(hipMemcpyAsync(w->d_text[s].p, b->text.p + text_lo, text_hi - text_lo, hipMemcpyHostToDevice, stream));
                if (uniform) {
                    seg[s].bases = reinterpret_cast<const uint8_t*>(w->d_text[s].p) + (r[0].seq_start() - text_lo);
                    seg[s].uniform_len = r[0].seq_len; seg[s].uniform_stride = static_cast<uint32_t>(stride);
                } else {
                    w->off[s].reserve(n); w->len[s].reserve(n); w->d_off[s].reserve(n); w->d_len[s].reserve(n);
                    for (size_t k = 0; k < n; ++k) { w->off[s].p[k] = r[k].seq_start() - text_lo; w->len[s].p[k] = r[k].seq_len; }
                    HIP_OK(hipMemcpyAsync(w->d_off[s].p, w->off[s].p, n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                    HIP_OK(hipMemcpyAsync(w->d_len[s].p, w->len[s].p, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                    seg[s].bases = reinterpret_cast<const uint8_t*>(w->d_text[s].p);
                    seg[s].offsets = w->d_off[s].p; seg[s].lengths = w->d_len[s].p;
                }
                side[s].pos += n;
            }
            // the previous batch must be complete before this one's flags can be trusted (and
            // its scan overlapped the GPU work): finish it first, then launch
            if (inflight) { Work* p = inflight; inflight = nullptr; finish(p); if (bad_base) { for (int s = 0; s < S; ++s) w->blk[s]->release(); free_works.push(w); break; } }
            const int rc = fqd_submit(eng.e, seg, n, FQD_MEM_DEVICE, w->d_keep.p);
            if (rc != FQD_OK) { for (int s = 0; s < S; ++s) w->blk[s]->release(); free_works.push(w);
                                throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.e)); }
            HIP_OK(hipMemcpyAsync(w->keep.p, w->d_keep.p, n, hipMemcpyDeviceToHost, stream));
            inflight = w;
            next_index += n;
        }
        if (inflight) { Work* p = inflight; inflight = nullptr; finish(p); }
    } catch (...) {
        if (inflight) { (void)hipStreamSynchronize(stream); for (int s = 0; s < S; ++s) inflight->blk[s]->release(); free_works.push(inflight); }
        stop_writer();
        throw;
    }
    { StageClock::Scope t("main: drain writers"); stop_writer(); }
    writers.rethrow();
    { StageClock::Scope t("main: close outputs"); for (int s = 0; s < S; ++s) sink[s]->close(); }
    StageClock::report();

    fqd_stats st{};
    fqd_get_stats(eng.e, &st);
    if (bad_base) throw_unknown_base(bad_byte);                // partial output stays on disk, as in the reference

    // A malformed record is noticed by the one-record lookahead while the record before it is
    // being fetched; it only fires if that fetch happens, i.e. the other file still has a
    // record at this position (left file first: hpp:232-233).
    for (int s = 0; s < S; ++s) {
        if (side[s].available() == 0 && side[s].failed && side[s].held_back) {
            bool other_has = true;
            if (S == 2) other_has = side[1 - s].has_record_here();
            if (other_has) { std::cerr << side[s].failure.diag; throw std::runtime_error(side[s].failure.what); }
        }
    }

    summary_.total = next_index; summary_.duplicates = st.duplicates; summary_.unmatched = 0;
    if (verbose_) {
        if (S == 1) std::cout << summary_.total << " reads processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        else        std::cout << summary_.total << " read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
    }
}

void HashDupRemover::filterSE(const std::string& infile, const std::string& outfile)
{
    try { if (tuning_.devices.empty()) { if (!run_ordered_resident(1, &infile, &outfile)) run_ordered(1, &infile, &outfile); } else run_ordered_multi(1, &infile, &outfile); }
    catch (const DiagnosedError& e) { std::cerr << e.diag; throw; }
}

void HashDupRemover::filterPE(const std::string& infile1, const std::string& infile2,
                              const std::string& outfile1, const std::string& outfile2, bool unordered)
{
    const std::string in[2] = {infile1, infile2}, out[2] = {outfile1, outfile2};
    try {
        if (unordered) run_unordered(in, out);
        else if (tuning_.devices.empty()) { if (!run_ordered_resident(2, in, out)) run_ordered(2, in, out); }
        else           run_ordered_multi(2, in, out);
    } catch (const DiagnosedError& e) { std::cerr << e.diag; throw; }
}

} // namespace fqdhost
