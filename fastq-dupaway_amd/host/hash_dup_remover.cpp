#include "hash_dup_remover.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <filesystem>
#include <hip/hip_runtime_api.h>
#include <iostream>
#include <memory>
#include <mutex>
#include <random>
#include <thread>

#include "fqdupaway.h"
#include "id_join.hpp"
#include "multi_gpu.hpp"

namespace fqdhost {

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

namespace {

// What went wrong on the GPU side, told apart so that a run that can hand over to another way of running knows why it
// does: out of HBM (hand over silently: the other way needs less) or a device / engine error (hand over, but SAY so).
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct DeviceOutOfMemory : DeviceError { using DeviceError::DeviceError; };

#define HIP_OK(expr)                                                                        \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError();           /* not sticky: a fallback may follow */ \
        if (e_ == hipErrorOutOfMemory) throw DeviceOutOfMemory(std::string(#expr) + ": " + hipGetErrorString(e_)); \
        throw DeviceError(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

// One line on stderr when a GPU-resident run gives up because of a DEVICE ERROR (not because it does not apply, and not
// for lack of HBM) and the streaming run takes over: the user learns that the fast path failed and on which call.
void announce_handover(const char* run, const std::exception& e)
{
    std::cerr << "[fastq-dupaway] " << run << " gave up on a GPU error (" << e.what() << "); continuing with the streaming run\n";
}

// RAII over the C ABI
// Set by a resident run of the CLI after its outputs are closed (Tuning::leave_memory_to_exit): from then on
// buffers are not freed one by one — the process is about to end and the driver releases everything at once.
static std::atomic<bool> g_leave_memory_to_exit{false};

struct EngineHandle {
    fqd_engine* e = nullptr;
    EngineHandle(int segments, int device, hipStream_t stream, uint64_t capacity_reads = 0, uint64_t capacity_bases = 0)
    {
        fqd_config cfg{};
        cfg.device = device; cfg.segments = segments; cfg.stream = stream;
        cfg.capacity_reads = capacity_reads; cfg.capacity_bases = capacity_bases;
        const int rc = fqd_engine_create(&cfg, &e);
        if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(nullptr));
    }
    ~EngineHandle() { if (!g_leave_memory_to_exit) fqd_engine_destroy(e); }
};

// The reference's two lines for a byte outside {A,C,G,T,N} (seq_utils.cpp:17-19).
[[noreturn]] void throw_unknown_base(uint8_t byte)
{
    std::cerr << "Error: unknown character in DNA sequence: " << static_cast<char>(byte) << '\n';
    throw std::runtime_error("Supported sequence character set: {A, N, C, G, T}!");
}

template <class T>
struct Pinned {
    T* p = nullptr; size_t cap = 0;
    ~Pinned() { if (p && !g_leave_memory_to_exit) (void)hipHostFree(p); }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        void* np = nullptr;
        HIP_OK(hipHostMalloc(&np, std::max<size_t>(n, 1024) * sizeof(T), hipHostMallocPortable));
        p = static_cast<T*>(np); cap = std::max<size_t>(n, 1024);
    }
};

template <class T>
struct Device {
    T* p = nullptr; size_t cap = 0;
    ~Device() { if (p && !g_leave_memory_to_exit) (void)hipFree(p); }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        void* np = nullptr;
        HIP_OK(hipMalloc(&np, std::max<size_t>(n, 1024) * sizeof(T)));
        p = static_cast<T*>(np); cap = std::max<size_t>(n, 1024);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Thread-safe pool / queue of raw pointers.
template <class T>
class Channel {
public:
    void push(T* v) { { std::lock_guard<std::mutex> g(m_); q_.push_back(v); } cv_.notify_one(); }
    T* pop() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); T* v = q_.front(); q_.pop_front(); return v; }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T*> q_;
};

struct PooledBlock : Block {
    Channel<PooledBlock>* home = nullptr;
    std::atomic<int> users{0};
    std::exception_ptr error;        // the reader thread's fill() threw: rethrown by the consumer
    bool stream_end = false;         // no block: the stream had already ended
    void acquire() { users.fetch_add(1); }
    void release() { if (users.fetch_sub(1) == 1) home->push(this); }
};

// One side (file) of an ordered run: a reader thread fills pinned blocks (file read + record
// scan) ahead of the consumer, which walks them with a cursor.
struct Side {
    std::unique_ptr<RecordStream> stream;
    Channel<PooledBlock> pool, raw, ready;
    std::vector<std::unique_ptr<PooledBlock>> storage;
    std::thread reader, scanner;
    std::atomic<bool> stop{false};
    int device = 0;
    PooledBlock* cur = nullptr;      // block being consumed (holds one "feeder" reference)
    size_t pos = 0;                  // next record of cur
    bool ended = false;              // no further records will come
    bool failed = false; ParseFailure failure; bool held_back = false;

    ~Side() { shutdown(); }
    void open_file(const std::string& name, Format f, bool want_tag, size_t block_bytes)
    {
        stream = std::make_unique<RecordStream>(name, f, want_tag, block_bytes);
    }
    void prime(int n_blocks, int dev)
    {
        device = dev;
        for (int k = 0; k < n_blocks; ++k) {
            storage.emplace_back(new PooledBlock());
            storage.back()->home = &pool;
            pool.push(storage.back().get());
        }
        // two stages on two threads: `reader` fetches block k+1 from the file while `scanner` scans block k
        reader = std::thread([this] {
            (void)hipSetDevice(device);
            for (;;) {
                PooledBlock* b;
                { StageClock::Scope t("reader: wait for a free block"); b = pool.pop(); }
                if (!b || stop.load()) break;
                b->error = nullptr; b->stream_end = false;
                bool more = false;
                try { more = stream->read_raw(*b); }
                catch (...) { b->error = std::current_exception(); raw.push(b); break; }
                if (!more) { b->stream_end = true; raw.push(b); break; }
                const bool last = b->raw_eof;
                raw.push(b);
                if (last) break;
            }
        });
        scanner = std::thread([this] {
            (void)hipSetDevice(device);
            for (;;) {
                PooledBlock* b;
                { StageClock::Scope t("scanner: wait for a raw block"); b = raw.pop(); }
                if (!b) break;
                if (b->error || b->stream_end) { ready.push(b); break; }
                try { stream->finish(*b); }
                catch (...) { b->error = std::current_exception(); ready.push(b); break; }
                const bool last = b->last;
                ready.push(b);
                if (last) break;
            }
        });
        advance();                   // the reference parses the first record when the file is set (bufferedinput.hpp:38-42)
    }
    void shutdown()
    {
        if (reader.joinable()) { stop.store(true); pool.push(nullptr); reader.join(); }
        if (scanner.joinable()) { raw.push(nullptr); scanner.join(); }
    }
    // Makes `cur` a block with unread records, or marks the side ended.
    void advance()
    {
        while (!ended && (cur == nullptr || pos >= cur->recs.size())) {
            if (cur) {
                const bool was_last = cur->last;
                if (cur->failure.set) { failed = true; failure = cur->failure; held_back = cur->held_back; }
                cur->release(); cur = nullptr;
                if (was_last) { ended = true; break; }
            }
            PooledBlock* b;
            { StageClock::Scope t("main: wait for a block"); b = ready.pop(); }
            if (b->error) { std::exception_ptr err = b->error; b->error = nullptr; pool.push(b); ended = true; std::rethrow_exception(err); }
            if (b->stream_end) { pool.push(b); ended = true; break; }
            b->users.store(1);       // the feeder's reference
            cur = b; pos = 0;
        }
    }
    size_t available() { advance(); return ended ? 0 : cur->recs.size() - pos; }
    // Does a fetched-but-unprocessed record sit at the current position?  (see Block::held_back)
    bool has_record_here() { return available() > 0 || (failed && held_back); }
};

// One batch travelling feeder -> GPU -> writer.
struct Work {
    int S = 1;
    PooledBlock* blk[2] = {nullptr, nullptr};
    size_t begin[2] = {0, 0};
    size_t n = 0;
    uint64_t first_index = 0;        // pair index of the batch's first record
    uint64_t emit_below = ~0ull;     // records at or beyond this pair index are not written
    bool stop = false;               // tells the writers to finish
    std::atomic<int> writers_left{0}; // one writer thread per output file
    Channel<Work>* home = nullptr;   // pool the batch goes back to (multi-GPU runs: one pool per rank)
    Pinned<uint64_t> off[2]; Pinned<uint32_t> len[2]; Pinned<uint8_t> keep;
    Device<char> d_text[2]; Device<uint64_t> d_off[2]; Device<uint32_t> d_len[2]; Device<uint8_t> d_keep;
};

} // namespace

namespace {

// Writer threads (one per output file: gzip outputs deflate in parallel): survivors, verbatim, in
// the order the batches are handed over.
unsigned write_threads()
{
    static const unsigned t = [] { const char* v = std::getenv("FQD_WRITE_THREADS"); const int x = v ? std::atoi(v) : 0; return x > 0 ? unsigned(x) : std::min(8u, host_threads()); }();
    return t;
}

class SurvivorWriters {
public:
    SurvivorWriters(int S, std::unique_ptr<OutputFile>* sinks, Channel<Work>* recycle) : S_(S), sink_(sinks), recycle_(recycle)
    {
        for (int s = 0; s < S_; ++s) thread_[s] = std::thread([this, s] { body(s); });
    }
    ~SurvivorWriters()
    {
        if (!thread_[0].joinable()) return;                    // stopped the regular way
        last_resort_.stop = true; last_resort_.home = &nowhere_;   // an exception is unwinding past us: end the threads
        hand_over(&last_resort_);
        for (int s = 0; s < S_; ++s) thread_[s].join();
    }
    void hand_over(Work* w) { w->writers_left.store(S_); for (int s = 0; s < S_; ++s) queue_[s].push(w); }
    // `w`: a free Work used as the stop marker.
    void stop(Work* w) { w->stop = true; hand_over(w); for (int s = 0; s < S_; ++s) thread_[s].join(); }
    void rethrow() { for (int s = 0; s < S_; ++s) if (error_[s]) std::rethrow_exception(error_[s]); }
private:
    void body(int s)
    {
        bool failed_already = false;
        std::vector<OutputFile::Piece> pieces;
        for (;;) {
            Work* w;
            { StageClock::Scope t("writer: wait for a batch"); w = queue_[s].pop(); }
            const bool stop = w->stop;
            if (!stop && !failed_already) {
                StageClock::Scope t("writer: write survivors");
                try {
                    const Block& b = *w->blk[s];
                    pieces.clear();                          // runs of adjacent survivors, written where they lie
                    const char* run_from = nullptr; size_t run_len = 0;
                    for (size_t k = 0; k < w->n; ++k) {
                        const RecordRef& r = b.recs[w->begin[s] + k];
                        const bool keep = w->keep.p[k] != 0 && w->first_index + k < w->emit_below;
                        if (keep) {
                            const char* p = b.text.p + r.start;
                            if (run_from && run_from + run_len == p) run_len += r.size;
                            else { if (run_len) pieces.push_back({run_from, run_len}); run_from = p; run_len = r.size; }
                        }
                    }
                    if (run_len) pieces.push_back({run_from, run_len});
                    StageClock::Scope t2("writer: copy out");
                    sink_[s]->write_pieces(pieces.data(), pieces.size(), write_threads());
                } catch (...) { error_[s] = std::current_exception(); failed_already = true; }
            }
            if (!stop) w->blk[s]->release();
            Channel<Work>* home = w->home ? w->home : recycle_;
            if (w->writers_left.fetch_sub(1) == 1) home->push(w);
            if (stop) break;
        }
    }
    int S_; std::unique_ptr<OutputFile>* sink_; Channel<Work>* recycle_;
    Channel<Work> queue_[2]; std::thread thread_[2]; std::exception_ptr error_[2];
    Work last_resort_; Channel<Work> nowhere_;
};

} // namespace

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

    try {
        while (!bad_base) {
            // ---- deal the next batches to the ranks, in file order ----------------------------------
            round.assign(size_t(N), nullptr);
            std::vector<fqd_reads> seg(size_t(N) * size_t(S));
            std::vector<uint64_t> n_of(size_t(N), 0);
            std::vector<uint8_t*> keep_of(size_t(N), nullptr);
            bool any = false;
            for (int r = 0; r < N; ++r) {
                size_t n = round_reads;
                for (int s = 0; s < S; ++s) n = std::min(n, side[s].available());
                if (n == 0) break;
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
                    const uint64_t stride = n > 1 ? rr[1].seq_start() - rr[0].seq_start() : rr[0].size;
                    bool uniform = stride <= 0xFFFFFFFFull && rr[0].seq_len > 0;
                    uint32_t longest = rr[0].seq_len;
                    for (size_t i = 1; i < n; ++i) {
                        uniform = uniform && rr[i].seq_len == rr[0].seq_len && rr[i].seq_start() - rr[i - 1].seq_start() == stride;
                        longest = std::max(longest, rr[i].seq_len);
                    }
                    // what the group was made for: one fixed length per mate, or (padded keys) anything up to a maximum
                    const char* refuse = nullptr;
                    if (shard.g && !padded && (!uniform || rr[0].seq_len != (s ? len1 : len0)))
                        refuse = "FQD_DEVICES: the first blocks held reads of one fixed length, a later one does not: run again with FQD_SHARD_PADDED=1";
                    if (shard.g && padded && longest > (s ? len1 : len0))
                        refuse = "FQD_DEVICES: a read longer than any in the first blocks turned up: run again with FQD_SHARD_MAX_LEN=<longest read>";
                    if (refuse) {
                        for (int q = 0; q <= s; ++q) w->blk[q]->release();
                        k.pool.push(w);
                        throw std::runtime_error(refuse);
                    }
                    if (!shard.g) {
                        all_uniform = all_uniform && uniform && (!have_shape_of[s] || rr[0].seq_len == (s ? len1 : len0));
                        have_shape_of[s] = true;
                        (s ? len1 : len0) = std::max(s ? len1 : len0, longest);
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
                size_t most = 0;
                for (uint64_t x : n_of) most = std::max<size_t>(most, size_t(x));
                round_reads = std::min(kMaxBatch, most + most / 4 + 1024);
                const char* force = std::getenv("FQD_SHARD_PADDED");
                padded = !all_uniform || (force && force[0] == '1');
                if (padded) {
                    // room to spare above the longest read of the first blocks: a whole 32-base group costs one key word
                    len0 = (len0 + 31u) / 32u * 32u; len1 = (len1 + 31u) / 32u * 32u;
                    if (const char* v = std::getenv("FQD_SHARD_MAX_LEN")) { const uint32_t x = uint32_t(std::strtoul(v, nullptr, 10)); len0 = std::max(len0, x); if (S == 2) len1 = std::max(len1, x); }
                    cfg.flags |= FQD_SHARD_PADDED;
                }
                cfg.round_reads = round_reads; cfg.len0 = len0; cfg.len1 = S == 2 ? len1 : 0;
                if (const char* v = std::getenv("FQD_SHARD_SLAB")) cfg.slab_records = std::strtoull(v, nullptr, 10);    // tests: force slab overflows
                if (tuning_.use_rccl) { if (fqd_shard_unique_id(id) != FQD_OK) throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr)); cfg.unique_id = id; }
                if (fqd_shard_create(engines.data(), &cfg, &shard.g) != FQD_OK)
                    throw std::runtime_error(std::string("GPU exchange: ") + fqd_shard_last_error(nullptr));
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

namespace {

// Copies entry k of a device array of uint32.
uint32_t peek_u32(const uint32_t* d, uint64_t k, hipStream_t s)
{
    uint32_t v = 0;
    HIP_OK(hipMemcpyAsync(&v, d + k, sizeof v, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return v;
}

// One file of an `--unordered` run as the device sees it: where every record's tag and sequence lie.
struct DeviceSide {
    const uint8_t*  tag_bytes = nullptr;  const uint64_t* tag_off = nullptr;  const uint32_t* tag_len = nullptr;
    const uint8_t*  seq_bytes = nullptr;  const uint64_t* seq_off = nullptr;  const uint32_t* seq_len = nullptr;
    uint64_t n = 0;
};

// The device stage both `--unordered` paths share: join on the ID tag, apply the reference's
// end-of-file rule (or the full join), dedup the pairs in tag order.  Leaves on the device the pair
// lists (tag order) and one keep flag per processed pair.
struct JoinedPairs {
    Device<uint32_t> perm[2], match[2], pair[2];
    Device<uint8_t>  keep;
    Device<uint64_t> seq_off[2]; Device<uint32_t> seq_len[2];     // the pairs' sequences in tag order (the dedup's input)
    // Room for about `reads` records per file, made ahead of time (on a helper thread, under the reads of the inputs):
    // these are gigabytes, and a hipMalloc that has to wait for the driver to clear pages costs the stage that meets it
    // tenths of a second.  A guess that is too small costs what it always cost.
    void prepare(uint64_t reads)
    {
        for (int s = 0; s < 2; ++s) { perm[s].reserve(reads); match[s].reserve(reads); pair[s].reserve(reads); seq_off[s].reserve(reads); seq_len[s].reserve(reads); }
        keep.reserve(reads);
    }
    uint64_t n_proc = 0;            // pairs the reference processes
    uint64_t unmatched = 0;
    uint64_t written_below = 0;     // pairs at or beyond this index are not written (unknown base)
    bool bad = false; uint8_t bad_byte = 0;
};

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
        rc = fqd_submit(e, seg, n, FQD_MEM_DEVICE, jp.keep.p + a);
    }
    if (rc == FQD_OK) rc = fqd_engine_sync(e);
    if (rc == FQD_ERR_BAD_BASE) {
        uint64_t rec; uint32_t sg2, pos;
        fqd_bad_base(e, &rec, &sg2, &pos, &jp.bad_byte);
        jp.bad = true; jp.written_below = std::min<uint64_t>(rec, n_proc);
    } else engine_ok(rc);
}

// (DeviceOutOfMemory, above: a device allocation failed while the inputs were still being read — the caller may fall
// back to a way of running that needs less HBM.)

// Device memory that grows and keeps its contents.
template <class T>
struct GrowDevice {
    T* p = nullptr; size_t cap = 0, used = 0;
    ~GrowDevice() { if (p && !g_leave_memory_to_exit) (void)hipFree(p); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = used = 0; }
    void room_for(size_t more, hipStream_t s)
    {
        if (used + more <= cap) return;
        const size_t want = std::max<size_t>(used + more, std::max<size_t>(cap + cap / 2, 1u << 20));
        void* np = nullptr;
        if (hipMalloc(&np, want * sizeof(T) + 64) != hipSuccess) {
            (void)hipGetLastError();
            throw DeviceOutOfMemory("--unordered: the inputs do not fit in GPU memory (" + std::to_string((want * sizeof(T)) >> 20) + " MiB more needed)");
        }
        if (used) HIP_OK(hipMemcpyAsync(np, p, used * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
        if (p) (void)hipFree(p);
        p = static_cast<T*>(np); cap = want;
    }
};

bool is_regular_file(const std::string& name, uint64_t& size)
{
    std::error_code ec;
    const auto st = std::filesystem::status(name, ec);
    if (ec || !std::filesystem::is_regular_file(st)) return false;
    size = std::filesystem::file_size(name, ec);
    return !ec;
}

} // namespace

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

// One pass, text resident in HBM (see run_unordered).
// A file whose text stays in HBM: the text and, per record, where it starts, where its sequence starts, the
// lengths of its ID line and sequence, its size.
struct FileOnDevice {
    GrowDevice<char> text;
    GrowDevice<uint64_t> start, seq_off; GrowDevice<uint32_t> id_len, seq_len, size;
    Device<uint64_t> tag_off; Device<uint32_t> tag_len;
    uint64_t n = 0;
    void forget() { text.used = start.used = seq_off.used = id_len.used = seq_len.used = size.used = 0; n = 0; }
    void release() { text.release(); start.release(); seq_off.release(); id_len.release(); seq_len.release(); size.release(); tag_off.release(); tag_len.release(); n = 0; }
};

// A BGZF input of the resident run goes to HBM as it lies on disk — a fifth of its text — and is inflated and cut
// into records THERE (fqd_bgzf_inflate, fqd_scan_records): the host only reads the file and walks the member
// headers.  Whatever is not a regular, well-formed BGZF file holding whole records is read the host way instead
// (RecordStream), which is also what produces every diagnostic.  FQD_GUNZIP_DEVICE=0 turns it off.
struct CompressedOnDevice {
    Device<char> bytes;
    std::vector<uint64_t> comp_off, out_off;
    std::vector<uint32_t> comp_len, out_len, crc;
    uint64_t text_bytes = 0;
    bool inflated = false;                    // fetch_bgzf already inflated the members into the file's text (batch by batch, under the read)
    uint64_t bad_members = 0;
};

static bool inflate_on_device()
{
    const char* v = std::getenv("FQD_GUNZIP_DEVICE");
    return !v || std::atoi(v) != 0;
}

// false: not such a file (nothing is reported; the caller reads it the host way).
// `into` given: the members are inflated into into->text batch by batch WHILE the file is still being read (a batch =
// a few rounds of the chip's waves, one member each: fqd_bgzf_inflate_async on a small engine of this thread's own), and
// the room for the text — sized from the file's size before anything is known about its members, regrown if that was
// too little — is allocated by a helper thread under the first reads: on a device whose free memory another process
// has just given back, hipMalloc clears tens of gigabytes of pages and takes seconds (0.36 - 3.07 s of
// configs[4]'s wall).
static bool fetch_bgzf(const std::string& name, size_t block_bytes, int device, CompressedOnDevice& c, FileOnDevice* into = nullptr)
{
    uint64_t size = 0;
    if (!has_gz_extension(name) || !is_regular_file(name, size) || size < 28) return false;
    InputFile file(name, true);
    HIP_OK(hipSetDevice(device));
    hipStream_t up = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
    // two blocks: the copy of one to HBM runs under the read of the next (its event is waited for before the block is read into again)
    Pinned<char> blocks[2];
    hipEvent_t sent[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    struct SentGuard { hipEvent_t* e; ~SentGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } sent_guard{sent};
    for (int k = 0; k < 2; ++k) { blocks[k].reserve(block_bytes); HIP_OK(hipEventCreateWithFlags(&sent[k], hipEventDisableTiming)); }
    c.bytes.reserve(size + 64);
    // ---- inflate under the read ---------------------------------------------------------------------------------
    static const bool overlap = [] { const char* v = std::getenv("FQD_INFLATE_OVERLAP"); return !v || std::atoi(v) != 0; }();
    const bool inflating = into != nullptr && overlap;
    constexpr uint64_t kBatchMembers = 32768;                 // eight rounds of the chip's 4096 waves, one member each: 2 GB of text, ~10 ms
    std::unique_ptr<EngineHandle> codec;                      // this thread's engine: scratch and stream of the inflate launches
    hipStream_t codec_stream = nullptr;
    struct CodecGuard { hipStream_t& s; std::unique_ptr<EngineHandle>& e; ~CodecGuard() { e.reset(); if (s) (void)hipStreamDestroy(s); } } cg{codec_stream, codec};
    std::thread room;                                          // allocates into->text
    std::exception_ptr room_error;
    struct RoomGuard { std::thread& t; ~RoomGuard() { if (t.joinable()) t.join(); } } rg{room};
    struct Batch { Device<uint64_t> comp_off, out_off; Device<uint32_t> comp_len, out_len, crc; };
    std::vector<std::unique_ptr<Batch>> batches;              // alive until the stream has drained
    Device<uint64_t> d_bad;
    uint64_t launched = 0, launched_bytes = 0;                // members / text bytes handed to the device so far
    hipEvent_t copied = nullptr;
    struct EventGuard { hipEvent_t& e; ~EventGuard() { if (e) (void)hipEventDestroy(e); } } eg{copied};
    if (inflating) {
        HIP_OK(hipStreamCreateWithFlags(&codec_stream, hipStreamNonBlocking));
        codec = std::make_unique<EngineHandle>(1, device, codec_stream);
        HIP_OK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        d_bad.reserve(2);
        HIP_OK(hipMemsetAsync(d_bad.p, 0, 2 * sizeof(uint64_t), codec_stream));
        const uint64_t guess = size * 6 + (64u << 20);        // level-1 FASTQ inflates 4-5.6x
        room = std::thread([&, guess] {
            try { HIP_OK(hipSetDevice(device)); StageClock::Scope t("  on the GPU: room for the text (under the read)"); into->text.room_for(guess, nullptr); }
            catch (...) { room_error = std::current_exception(); }
        });
    }
    auto launch_batch = [&](bool last) {
        const uint64_t have = c.comp_off.size();
        if (!inflating || have == launched || (!last && have - launched < kBatchMembers)) return;
        if (room.joinable()) { room.join(); if (room_error) std::rethrow_exception(room_error); }
        const uint64_t n = have - launched, need = c.text_bytes + 64;
        if (need > into->text.cap) {                           // the guess was too small: everything inflated so far moves
            HIP_OK(hipStreamSynchronize(codec_stream));
            into->text.used = launched_bytes;
            into->text.room_for(std::max<uint64_t>(need, into->text.cap + into->text.cap / 2) - into->text.used, codec_stream);
        }
        batches.emplace_back(new Batch());
        Batch& b = *batches.back();
        b.comp_off.reserve(n); b.out_off.reserve(n); b.comp_len.reserve(n); b.out_len.reserve(n); b.crc.reserve(n);
        // the member arrays go up on the COPY stream: waiting for them must not wait for the batch before this one
        HIP_OK(hipMemcpyAsync(b.comp_off.p, c.comp_off.data() + launched, n * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.out_off.p, c.out_off.data() + launched, n * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.comp_len.p, c.comp_len.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.out_len.p, c.out_len.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.crc.p, c.crc.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipStreamSynchronize(up));                      // (the vectors may grow and move under the next block's walk)
        HIP_OK(hipEventRecord(copied, up));                    // arrays and compressed bytes of these members are on the device
        HIP_OK(hipStreamWaitEvent(codec_stream, copied, 0));
        if (fqd_bgzf_inflate_async(codec->e, reinterpret_cast<const uint8_t*>(c.bytes.p), b.comp_off.p, b.comp_len.p, b.out_off.p, b.out_len.p,
                                   b.crc.p, n, reinterpret_cast<uint8_t*>(into->text.p), d_bad.p) != FQD_OK)
            throw DeviceError(std::string("GPU engine: ") + fqd_last_error(codec->e));
        launched = have; launched_bytes = c.text_bytes;
    };
    std::string tail;                          // bytes already read from `tail_at` on: a member may straddle two blocks
    uint64_t tail_at = 0, at = 0, member = 0;  // file offsets: of the tail, of the current block, of the member being parsed
    for (int turn = 0;; turn ^= 1) {
        Pinned<char>& block = blocks[turn];
        if (in_flight[turn]) { HIP_OK(hipEventSynchronize(sent[turn])); in_flight[turn] = false; }
        const size_t got = file.read(block.p, block_bytes, host_threads());
        if (got == 0) break;
        if (at + got > size) { (void)hipStreamSynchronize(up); return false; }     // the file grew under us
        HIP_OK(hipMemcpyAsync(c.bytes.p + at, block.p, got, hipMemcpyHostToDevice, up));
        HIP_OK(hipEventRecord(sent[turn], up)); in_flight[turn] = true;
        auto fetch = [&](uint64_t from, size_t len, unsigned char* dst) {
            if (from + len > at + got) return false;
            for (size_t k = 0; k < len; ++k)
                dst[k] = static_cast<unsigned char>(from + k >= at ? block.p[from + k - at] : tail[from + k - tail_at]);
            return true;
        };
        bool ok = true;
        for (;;) {
            unsigned char head[18], trailer[8];
            if (!fetch(member, sizeof head, head)) break;
            size_t data_off = 0;
            const size_t total = bgzf_member_size(head, sizeof head, &data_off);
            if (total == 0) { ok = false; break; }                     // not BGZF (or an extra field of another shape)
            if (!fetch(member + total - 8, sizeof trailer, trailer)) break;
            const uint32_t crc = trailer[0] | (uint32_t(trailer[1]) << 8) | (uint32_t(trailer[2]) << 16) | (uint32_t(trailer[3]) << 24);
            const uint32_t isize = trailer[4] | (uint32_t(trailer[5]) << 8) | (uint32_t(trailer[6]) << 16) | (uint32_t(trailer[7]) << 24);
            if (isize > 65536u) { ok = false; break; }
            if (isize) {
                c.comp_off.push_back(member + data_off); c.comp_len.push_back(static_cast<uint32_t>(total - data_off - 8));
                c.out_off.push_back(c.text_bytes); c.out_len.push_back(isize); c.crc.push_back(crc);
                c.text_bytes += isize;
            }
            member += total;
        }
        if (!ok) { (void)hipStreamSynchronize(up); if (inflating) (void)hipStreamSynchronize(codec_stream); return false; }
        launch_batch(false);
        std::string keep;
        if (member < at + got) {
            if (member < at) keep.assign(tail, static_cast<size_t>(member - tail_at), std::string::npos);
            const uint64_t from = std::max(member, at);
            keep.append(block.p + (from - at), static_cast<size_t>(at + got - from));
        }
        tail.swap(keep); tail_at = member;
        at += got;
    }
    HIP_OK(hipStreamSynchronize(up));                          // every byte of the file is in HBM
    const bool whole = at == size && member == size && c.text_bytes > 0;
    if (inflating) {
        if (whole) launch_batch(true);
        HIP_OK(hipStreamSynchronize(codec_stream));
        if (room.joinable()) { room.join(); if (room_error) std::rethrow_exception(room_error); }
        if (whole) {
            uint64_t bad[2] = {0, 0};
            HIP_OK(hipMemcpy(bad, d_bad.p, sizeof bad, hipMemcpyDeviceToHost));
            c.bad_members = bad[0] + bad[1];
            c.inflated = true;
            StageClock::Scope t("  on the GPU: compressed bytes freed");
            c.bytes.release();
        }
    }
    return whole;
}

static bool records_on_device(fqd_engine* e, hipStream_t stream, Format format, uint64_t text_bytes, FileOnDevice& f);

// The GPU's share of a file that arrived compressed: inflate, count lines, cut into records.  false: a damaged
// member, or text that is not whole records — the caller reads the file the host way, which says what is wrong.
static bool finish_on_device(fqd_engine* e, hipStream_t stream, Format format, CompressedOnDevice& c, FileOnDevice& f)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    if (c.inflated) {                                         // fetch_bgzf did it under the read
        if (c.bad_members) return false;
        return records_on_device(e, stream, format, c.text_bytes, f);
    }
    const uint64_t members = c.comp_off.size();
    Device<uint64_t> d_comp_off, d_out_off; Device<uint32_t> d_comp_len, d_out_len, d_crc;
    d_comp_off.reserve(members); d_out_off.reserve(members); d_comp_len.reserve(members); d_out_len.reserve(members); d_crc.reserve(members);
    HIP_OK(hipMemcpy(d_comp_off.p, c.comp_off.data(), members * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_out_off.p, c.out_off.data(), members * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_comp_len.p, c.comp_len.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_out_len.p, c.out_len.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_crc.p, c.crc.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    { StageClock::Scope t("  on the GPU: room for the text"); f.text.room_for(c.text_bytes + 64, stream); }
    uint64_t bad = 0;
    {
        StageClock::Scope t("  on the GPU: inflate + CRC check");
        engine_ok(fqd_bgzf_inflate(e, reinterpret_cast<const uint8_t*>(c.bytes.p), d_comp_off.p, d_comp_len.p, d_out_off.p, d_out_len.p,
                                   d_crc.p, members, reinterpret_cast<uint8_t*>(f.text.p), &bad));
    }
    { StageClock::Scope t("  on the GPU: compressed bytes freed"); c.bytes.release(); }
    if (bad) return false;
    return records_on_device(e, stream, format, c.text_bytes, f);
}

// The text of a file is in HBM (f.text, text_bytes of it): cut it into records there.  false: not whole records.
static bool records_on_device(fqd_engine* e, hipStream_t stream, Format format, uint64_t text_bytes, FileOnDevice& f)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    struct { uint64_t text_bytes; } c{text_bytes};
    const uint32_t lines_per_record = format == Format::Fastq ? 4u : 2u;
    uint64_t lines = 0;
    { StageClock::Scope t("  on the GPU: line count"); engine_ok(fqd_count_lines(e, reinterpret_cast<const uint8_t*>(f.text.p), c.text_bytes, &lines)); }
    const uint64_t n = lines / lines_per_record;
    {
        StageClock::Scope t("  on the GPU: room for the record arrays");
        f.start.room_for(n, stream); f.seq_off.room_for(n, stream); f.id_len.room_for(n, stream); f.seq_len.room_for(n, stream); f.size.room_for(n, stream);
    }
    int well_formed = 0;
    {
        StageClock::Scope t("  on the GPU: record scan");
        engine_ok(fqd_scan_records(e, reinterpret_cast<const uint8_t*>(f.text.p), c.text_bytes, lines_per_record, n,
                                   f.start.p, f.seq_off.p, f.id_len.p, f.seq_len.p, f.size.p, &well_formed));
    }
    if (!well_formed || n == 0) return false;
    f.text.used = c.text_bytes;
    f.start.used = f.seq_off.used = f.id_len.used = f.seq_len.used = f.size.used = n;
    f.n = n;
    return true;
}

// `.gz` outputs of the resident run: deflated on the GPU (fqd_bgzf_deflate; the size of zlib level 1-2 at a
// small fraction of its time) unless a level was asked for — FQD_GZ_LEVEL=N means the host codec at level N —
// or FQD_GZ_DEVICE=0/1 says otherwise.
static bool deflate_on_device()
{
    if (const char* v = std::getenv("FQD_GZ_DEVICE")) return std::atoi(v) != 0;
    return std::getenv("FQD_GZ_LEVEL") == nullptr;
}

// The outputs of a run whose text is in HBM: pair k < upto (record idx[s][k] of file s; idx[s] == nullptr: record k)
// is written iff keep[k].  The device assembles windows of survivors in output order (and deflates them, for `.gz`
// outputs: deflate_on_device), the host writes what comes back, a writer thread per file.  Closes the sinks.
// Everything write_survivors allocates — the output plan of each file, the window and member buffers on the device,
// the pinned buffers the windows come back in — so that a run can have it all BEFORE it creates an output (ADVICE r2:
// an allocation that fails after the sinks exist leaves truncated files and no way back to the streaming run).
struct SurvivorBuffers {
    struct PerFile {
        Device<uint64_t> src_off, dst_off; Device<uint32_t> len; uint64_t total = 0;
        Pinned<char> buf[2]; Device<char> d_win[2], d_members[2];      // a slot = pinned buffer k + the device buffers k
        bool on_device = false;                       // .gz: windows leave the device as finished BGZF members
    } f[2];
    uint64_t window = 0, roomy = 0;
    bool planned = false;
};

static void plan_survivors(fqd_engine* e, int S, FileOnDevice* const* file, const uint32_t* const* idx, const uint8_t* keep, uint64_t upto,
                           const bool* gz_out, long long memlimit, SurvivorBuffers& b)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw DeviceError(std::string("GPU engine: ") + fqd_last_error(e)); };
    b.window = std::max<uint64_t>(4u << 20, static_cast<uint64_t>(memlimit > 0 ? memlimit : (2ll << 30)) / 16);   // bytes per buffer, two per file
    if (const char* v = std::getenv("FQD_STREAM_WINDOW_KB")) { const long kb = std::atol(v); if (kb > 0) b.window = static_cast<uint64_t>(kb) << 10; }
    b.roomy = b.window + b.window / 4;                      // the most a window may hold
    for (int s = 0; s < S; ++s) {
        SurvivorBuffers::PerFile& o = b.f[s];
        o.src_off.reserve(upto); o.dst_off.reserve(upto + 1); o.len.reserve(upto);
        engine_ok(fqd_output_plan(e, keep, idx[s], upto, file[s]->start.p, file[s]->size.p, o.src_off.p, o.len.p, o.dst_off.p, &o.total));
        o.on_device = gz_out[s] && deflate_on_device();
        // every buffer is sized once, for the largest window the writer lets through (a single record larger than that is the
        // one case that grows them later): a window a little larger than all before it must not cost a new pinned allocation
        const uint64_t room = std::min<uint64_t>(b.roomy, std::max<uint64_t>(o.total, 1));
        for (int k = 0; k < 2; ++k) {
            o.d_win[k].reserve(room + 64);
            o.buf[k].reserve((o.on_device ? std::max<uint64_t>(room / 2, 1u << 20) : room) + 64);
            if (o.on_device) o.d_members[k].reserve(fqd_bgzf_bound(room));
        }
    }
    b.planned = true;
}

static void write_survivors(fqd_engine* e, hipStream_t stream, int S, FileOnDevice* const* file, const uint32_t* const* idx,
                            const uint8_t* keep, uint64_t upto, uint64_t dups, OutputFile* const* sinks, Format format, long long memlimit,
                            bool close_sinks = true, SurvivorBuffers* planned = nullptr)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    SurvivorBuffers own;
    if (!planned || !planned->planned) {
        bool gz_out[2] = {false, false};
        for (int s = 0; s < S; ++s) gz_out[s] = sinks[s]->is_gz();
        plan_survivors(e, S, file, idx, keep, upto, gz_out, memlimit, own);
        planned = &own;
    }
    const uint64_t window = planned->window, roomy = planned->roomy;
    const uint32_t lines_per_record = format == Format::Fastq ? 4u : 2u;
    struct Out {
        Device<uint64_t>& src_off; Device<uint64_t>& dst_off; Device<uint32_t>& len; uint64_t total;
        Pinned<char>* buf; Device<char>* d_win; Device<char>* d_members;
        Channel<int> free_bufs, full_bufs; int slot_id[2] = {0, 1}; size_t bytes[2] = {0, 0};
        hipEvent_t copied[2] = {nullptr, nullptr};        // the slot's window has reached its pinned buffer
        bool on_device = false;
        std::thread writer; std::exception_ptr error;
        explicit Out(SurvivorBuffers::PerFile& p) : src_off(p.src_off), dst_off(p.dst_off), len(p.len), total(p.total), buf(p.buf), d_win(p.d_win), d_members(p.d_members), on_device(p.on_device) {}
    };
    Out o[2] = {Out(planned->f[0]), Out(planned->f[1])};
    static int kStop = -1;
    // A window leaves the device on a stream of its own while the kernels of the next one run: the writer thread waits
    // for the copy, not this loop.  (A slot's device buffers are free again when its pinned buffer is: the writer gives
    // the slot back after it has written it.)
    hipStream_t down = nullptr;
    hipEvent_t made = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&made, hipEventDisableTiming));
    for (int s = 0; s < S; ++s) for (int k = 0; k < 2; ++k) HIP_OK(hipEventCreateWithFlags(&o[s].copied[k], hipEventDisableTiming));
    struct DownGuard { hipStream_t& d; hipEvent_t& m; Out* o; ~DownGuard() {
        if (d) { (void)hipStreamSynchronize(d); (void)hipStreamDestroy(d); }
        if (m) (void)hipEventDestroy(m);
        for (int s = 0; s < 2; ++s) for (int k = 0; k < 2; ++k) if (o[s].copied[k]) (void)hipEventDestroy(o[s].copied[k]);
    } } down_guard{down, made, o};
    for (int s = 0; s < S; ++s) {
        o[s].free_bufs.push(&o[s].slot_id[0]); o[s].free_bufs.push(&o[s].slot_id[1]);
        o[s].writer = std::thread([&, s] {
            for (;;) {
                int* id = o[s].full_bufs.pop();
                if (*id < 0) break;
                try {
                    { StageClock::Scope t("  survivors: writer waits for the window's copy"); HIP_OK(hipEventSynchronize(o[s].copied[*id])); }
                    StageClock::Scope t("  survivors: writer writes");
                    if (!o[s].error) {
                        if (o[s].on_device) sinks[s]->write_members(o[s].buf[*id].p, o[s].bytes[*id], write_threads());
                        else if (sinks[s]->is_gz()) sinks[s]->write_borrowed(o[s].buf[*id].p, o[s].bytes[*id]);
                        else {                                   // a plain file: the window in slices, copied in by several threads
                            constexpr size_t kSlices = 128;
                            OutputFile::Piece pieces[kSlices];
                            const size_t n = o[s].bytes[*id];
                            for (size_t k = 0; k < kSlices; ++k) { const size_t a = n / kSlices * k, b = k + 1 == kSlices ? n : n / kSlices * (k + 1); pieces[k] = {o[s].buf[*id].p + a, b - a}; }
                            sinks[s]->write_pieces(pieces, kSlices, write_threads());
                        }
                    }
                }
                catch (...) { o[s].error = std::current_exception(); }
                o[s].free_bufs.push(id);
            }
        });
    }
    auto peek_u64 = [&](const uint64_t* d, uint64_t k) {
        uint64_t v = 0;
        HIP_OK(hipMemcpyAsync(&v, d + k, sizeof v, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return v;
    };
    std::exception_ptr failure;
    try {
        uint64_t at[2] = {0, 0};
        while (at[0] < upto || (S == 2 && at[1] < upto)) {
            for (int s = 0; s < S; ++s) {
                if (at[s] >= upto) continue;
                // as many pairs as fill a window: from the average record size, halved until the bytes fit
                const uint64_t avg = std::max<uint64_t>(1, o[s].total / std::max<uint64_t>(1, upto - dups));
                uint64_t take = std::min<uint64_t>(upto - at[s], std::max<uint64_t>(1, window / avg));
                const uint64_t lo = peek_u64(o[s].dst_off.p, at[s]);
                uint64_t hi;
                for (;;) {
                    hi = at[s] + take == upto ? o[s].total : peek_u64(o[s].dst_off.p, at[s] + take);
                    if (hi - lo <= roomy || take == 1) break;
                    take = std::max<uint64_t>(1, take / 2);
                }
                const uint64_t bytes = hi - lo;
                if (bytes) {
                    int* id = nullptr;
                    { StageClock::Scope t("  survivors: the device waits for a free buffer"); id = o[s].free_bufs.pop(); }
                    StageClock::Scope t("  survivors: windows made on the device");
                    // every buffer is sized once, for the largest window the loop above lets through: a window a little
                    // larger than all before it must not cost a new pinned allocation (tens of milliseconds each)
                    const uint64_t room = std::max(bytes, std::min<uint64_t>(roomy, o[s].total));      // (a small output: what it needs)
                    Device<char>& d_win = o[s].d_win[*id];
                    d_win.reserve(room + 64);
                    o[s].buf[*id].reserve((o[s].on_device ? std::max<uint64_t>(room / 2, 1u << 20) : room) + 64);
                    // dst_off is absolute in the output: the window's buffer starts `lo` bytes in
                    engine_ok(fqd_copy_spans(e, reinterpret_cast<const uint8_t*>(file[s]->text.p), o[s].src_off.p + at[s], o[s].len.p + at[s], take,
                                             reinterpret_cast<uint8_t*>(d_win.p) - lo, o[s].dst_off.p + at[s]));
                    uint64_t out_bytes = bytes;
                    const char* from = d_win.p;
                    if (o[s].on_device) {
                        Device<char>& d_members = o[s].d_members[*id];
                        const uint64_t cap = fqd_bgzf_bound(room);
                        d_members.reserve(cap);
                        engine_ok(fqd_bgzf_deflate(e, reinterpret_cast<const uint8_t*>(d_win.p), bytes, lines_per_record,
                                                   reinterpret_cast<uint8_t*>(d_members.p), cap, &out_bytes));
                        from = d_members.p;
                        o[s].buf[*id].reserve(out_bytes + 64);           // (text that does not shrink to half)
                    }
                    HIP_OK(hipEventRecord(made, stream));
                    HIP_OK(hipStreamWaitEvent(down, made, 0));
                    HIP_OK(hipMemcpyAsync(o[s].buf[*id].p, from, out_bytes, hipMemcpyDeviceToHost, down));
                    HIP_OK(hipEventRecord(o[s].copied[*id], down));
                    o[s].bytes[*id] = out_bytes;
                    o[s].full_bufs.push(id);
                }
                at[s] += take;
            }
        }
    } catch (...) { failure = std::current_exception(); }
    for (int s = 0; s < S; ++s) { o[s].full_bufs.push(&kStop); o[s].writer.join(); }
    if (failure) std::rethrow_exception(failure);
    for (int s = 0; s < S; ++s) { if (o[s].error) std::rethrow_exception(o[s].error); if (close_sinks) sinks[s]->close(); }
}

// What the dedup engine of a resident run will hold, guessed from the sizes of the input files before anything of them has
// been read, so that its key store and table can be allocated — and their pages cleared by the driver — on a helper thread
// under the reads (the key store's first hipMalloc was 1.05 s of a 1.13 s "pair dedup" stage).  A guess
// that is too small costs what it always cost (the store grows); one too large costs HBM nobody else wants.
static void guess_capacity(int S, const std::string* in, uint64_t& reads, uint64_t& bases)
{
    reads = bases = 0;
    uint64_t text[2] = {0, 0};
    for (int s = 0; s < S; ++s) {
        uint64_t size = 0;
        if (!is_regular_file(in[s], size)) { reads = bases = 0; return; }
        text[s] = has_gz_extension(in[s]) ? size * 5 : size;
    }
    const uint64_t least = S == 2 ? std::min(text[0], text[1]) : text[0];
    reads = least / 280 + 1024;                                // a 150-base FASTQ record is ~316 bytes
    bases = (text[0] + text[1]) / 2 + 4096;                    // about half of FASTQ text is sequence
}

// A plain regular file as it is to the tail of f.text (a pinned block, parallel preads, H2D); false: not such a file.
static bool fetch_plain(const std::string& name, size_t block_bytes, int device, FileOnDevice& f, uint64_t& text_bytes)
{
    uint64_t size = 0;
    if (has_gz_extension(name) || !is_regular_file(name, size) || size == 0) return false;
    InputFile file(name, true);
    HIP_OK(hipSetDevice(device));
    hipStream_t up = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
    Pinned<char> block[2];
    block[0].reserve(block_bytes); block[1].reserve(block_bytes);
    f.text.room_for(size + 64, up);
    uint64_t at = 0;
    for (int k = 0;; k ^= 1) {                                   // the copy of one block overlaps the read of the next
        const size_t got = file.read(block[k].p, block_bytes, host_threads());
        HIP_OK(hipStreamSynchronize(up));                          // the other block's copy
        if (got == 0) break;
        if (at + got > size) return false;                         // the file grew under us
        HIP_OK(hipMemcpyAsync(f.text.p + at, block[k].p, got, hipMemcpyHostToDevice, up));
        at += got;
    }
    text_bytes = at;
    return at == size;
}

// An ordered run (single-end, or paired files read side by side) with a codec at either end — BGZF inputs, or `.gz`
// outputs of plain regular inputs: the files go to HBM as they lie on disk, are inflated (if compressed) and cut into
// records there, every read (pair) is deduplicated where it lies, and the
// survivors leave in input order window by window (deflated on the device for `.gz` outputs).  Taken only when
// everything is plain sailing — regular BGZF files of whole records, as many in file 2 as in file 1, no unknown
// base, everything fits in HBM; otherwise false is returned BEFORE any output is touched and the streaming run
// (run_ordered), which reproduces the reference's behaviour for every irregular input, does the job.
bool HashDupRemover::run_ordered_resident(int S, const std::string* in, const std::string* out)
{
    if (const char* v = std::getenv("FQD_ORDERED_RESIDENT")) if (std::atoi(v) == 0) return false;
    if (!inflate_on_device()) return false;
    // worth it when a codec is involved: a BGZF input, or a `.gz` output the GPU can deflate (plain files in and
    // out are better off in the streaming run, where reading, the GPU and writing overlap)
    bool any_gz_in = false, any_gz_out = false;
    for (int s = 0; s < S; ++s) {
        uint64_t size = 0;
        if (!is_regular_file(in[s], size)) return false;
        any_gz_in |= has_gz_extension(in[s]);
        any_gz_out |= has_gz_extension(out[s]);
    }
    if (!any_gz_in && !(any_gz_out && deflate_on_device())) return false;
    HIP_OK(hipSetDevice(tuning_.device));
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));
    // a file fetched whole is read in larger pieces than the streaming run's blocks: the parallel read of a piece needs 16 MB per thread
    const size_t fetch_bytes = std::max<size_t>(block_bytes, std::min<size_t>(64u << 20, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : (64 << 20))));
    FileOnDevice dev[2];
    Device<uint8_t> keep;
    uint64_t n = 0, dups = 0;
    std::unique_ptr<EngineHandle> eng;
    SurvivorBuffers buffers;
    std::thread make_engine; std::exception_ptr engine_error;
    struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } join_guard{make_engine};
    try {
        CompressedOnDevice packed[2];
        bool fetched[2] = {false, false};
        std::exception_ptr fetch_error[2];
        uint64_t plain_bytes[2] = {0, 0};
        // the engine — its key store and table sized from the files' sizes — is made on a helper thread under the reads
        uint64_t cap_reads = 0, cap_bases = 0;
        guess_capacity(S, in, cap_reads, cap_bases);
        make_engine = std::thread([&] {
            try { HIP_OK(hipSetDevice(tuning_.device)); StageClock::Scope t("  on the GPU: engine, key store, table (under the read)"); eng = std::make_unique<EngineHandle>(S, tuning_.device, stream, cap_reads, cap_bases); }
            catch (...) { engine_error = std::current_exception(); }
        });
        {
            StageClock::Scope t("ordered/resident: files to HBM");
            auto fetch = [&](int s) {
                try {
                    fetched[s] = has_gz_extension(in[s]) ? fetch_bgzf(in[s], fetch_bytes, tuning_.device, packed[s], &dev[s])
                                                         : fetch_plain(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                } catch (const DeviceOutOfMemory&) { fetched[s] = false; }
                catch (const DeviceError&) { fetched[s] = false; fetch_error[s] = std::current_exception(); }
                catch (const std::exception&) { fetched[s] = false; }      // the host reader will say what is wrong with the file
            };
            std::thread second;
            if (S == 2) second = std::thread(fetch, 1);
            fetch(0);
            if (S == 2) second.join();
        }
        if (make_engine.joinable()) make_engine.join();
        for (int s = 0; s < S; ++s) if (fetch_error[s]) std::rethrow_exception(fetch_error[s]);
        for (int s = 0; s < S; ++s) if (!fetched[s]) return false;
        if (engine_error) std::rethrow_exception(engine_error);
        {
            StageClock::Scope t("ordered/resident: inflate + record scan on the GPU");
            for (int s = 0; s < S; ++s) {
                const bool ok = has_gz_extension(in[s]) ? finish_on_device(eng->e, stream, format_, packed[s], dev[s])
                                                        : records_on_device(eng->e, stream, format_, plain_bytes[s], dev[s]);
                if (!ok) return false;
            }
        }
        if (S == 2 && dev[0].n != dev[1].n) return false;
        n = dev[0].n;
        StageClock::Scope t("ordered/resident: dedup on the GPU");
        keep.reserve(n);
        const size_t kBatch = 16u << 20;
        int rc = FQD_OK;
        for (size_t a = 0; a < n && rc == FQD_OK; a += kBatch) {
            fqd_reads seg[2] = {};
            for (int s = 0; s < S; ++s) {
                seg[s].bases = reinterpret_cast<const uint8_t*>(dev[s].text.p);
                seg[s].offsets = dev[s].seq_off.p + a; seg[s].lengths = dev[s].seq_len.p + a;
            }
            rc = fqd_submit(eng->e, seg, std::min<size_t>(kBatch, n - a), FQD_MEM_DEVICE, keep.p + a);
        }
        if (rc == FQD_OK) rc = fqd_engine_sync(eng->e);
        if (rc == FQD_ERR_BAD_BASE) return false;                 // the streaming run cuts the output where the reference does
        if (rc != FQD_OK) throw DeviceError(std::string("GPU engine: ") + fqd_last_error(eng->e));
        if (std::getenv("FQD_TEST_FAIL_RESIDENT")) throw DeviceError("GPU engine: forced by FQD_TEST_FAIL_RESIDENT");      // tests: the hand-over is announced
        fqd_stats st{};
        fqd_get_stats(eng->e, &st);
        dups = st.duplicates;
        // everything the writer needs is reserved HERE, while the run can still hand over: once an output exists it cannot
        bool gz_out[2] = {false, false};
        for (int s = 0; s < S; ++s) gz_out[s] = has_gz_extension(out[s]);
        FileOnDevice* files[2] = {&dev[0], &dev[1]};
        const uint32_t* idx[2] = {nullptr, nullptr};
        plan_survivors(eng->e, S, files, idx, keep.p, n, gz_out, memlimit_, buffers);
    } catch (const DeviceOutOfMemory&) {
        return false;                                             // HBM that does not suffice: the streaming run needs a few blocks of it only
    } catch (const DeviceError& e) {
        announce_handover("the GPU-resident ordered run", e);     // nothing has been written yet
        return false;
    } catch (const std::exception&) {
        return false;                                             // an input the host reader will report on in the reference's words
    }
    // from here on the run is this one's: outputs are created, filled and closed
    OutputFile sink0(out[0]);
    std::unique_ptr<OutputFile> sink1;
    if (S == 2) sink1 = std::make_unique<OutputFile>(out[1]);
    OutputFile* sinks[2] = {&sink0, sink1.get()};
    {
        StageClock::Scope t("ordered/resident: survivors out of HBM");
        FileOnDevice* files[2] = {&dev[0], &dev[1]};
        const uint32_t* idx[2] = {nullptr, nullptr};
        write_survivors(eng->e, stream, S, files, idx, keep.p, n, dups, sinks, format_, memlimit_, true, &buffers);
    }
    if (tuning_.leave_memory_to_exit) g_leave_memory_to_exit = true;
    StageClock::report();
    summary_.total = n; summary_.duplicates = dups; summary_.unmatched = 0;
    if (verbose_) {
        if (S == 1) std::cout << summary_.total << " reads processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        else        std::cout << summary_.total << " read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
    }
    return true;
}

void HashDupRemover::run_unordered_resident(const std::string* in, const std::string* out)
{
    HIP_OK(hipSetDevice(tuning_.device));
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    // the engine — key store and table sized from the files' sizes — is made on a helper thread under the reads of the
    // inputs: its first use comes after them (guess_capacity)
    JoinedPairs jp;                                           // (before `eng`: the helper thread makes room in it, and eng's destructor joins that thread first)
    struct LazyEngine {
        std::unique_ptr<EngineHandle> holder; std::thread maker; std::exception_ptr error;
        ~LazyEngine() { if (maker.joinable()) maker.join(); }
        fqd_engine* get() { if (maker.joinable()) maker.join(); if (error) std::rethrow_exception(error); return holder->e; }
    } eng;
    {
        uint64_t cap_reads = 0, cap_bases = 0;
        guess_capacity(2, in, cap_reads, cap_bases);
        eng.maker = std::thread([this, &eng, &jp, stream, cap_reads, cap_bases] {
            try {
                HIP_OK(hipSetDevice(tuning_.device));
                StageClock::Scope t("  on the GPU: engine, key store, table, join arrays (under the read)");
                eng.holder = std::make_unique<EngineHandle>(2, tuning_.device, stream, cap_reads, cap_bases);
                if (cap_reads) jp.prepare(cap_reads);
            }
            catch (...) { eng.error = std::current_exception(); }
        });
    }
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.get())); };
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));
    // a file fetched whole is read in larger pieces than the streaming run's blocks: the parallel read of a piece needs 16 MB per thread
    const size_t fetch_bytes = std::max<size_t>(block_bytes, std::min<size_t>(64u << 20, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : (64 << 20))));

    FileOnDevice dev[2];

    // ---- the one pass: every block to the tail of the file's text in HBM -------------------------------
    {
        StageClock::Scope t("unordered/resident: read, scan, text to HBM");
        // both files at the same time, each on its own thread and copy stream; what goes wrong is still
        // reported in the reference's order: everything about file 1 before anything about file 2 (hpp:161-173)
        std::exception_ptr err[2];
        ParseFailure parse_failure[2];
        auto load = [&](int s) {
            try {
                HIP_OK(hipSetDevice(tuning_.device));
                hipStream_t up = nullptr;
                HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
                struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
                Pinned<uint64_t> h_start, h_seq; Pinned<uint32_t> h_idl, h_sql, h_size;
                FileOnDevice& f = dev[s];
                uint64_t known = 0;
                if (is_regular_file(in[s], known) && !has_gz_extension(in[s])) f.text.room_for(known + 64, up);   // no regrowth for plain files
                Side side;
                side.open_file(in[s], format_, true, block_bytes);
                side.prime(3, tuning_.device);
                while (side.available() > 0) {
                    PooledBlock* b = side.cur;
                    const size_t from = side.pos, nb = b->recs.size() - from;
                    const RecordRef* r = &b->recs[from];
                    const uint64_t text_lo = r[0].start, bytes = r[nb - 1].start + r[nb - 1].size - text_lo;
                    f.text.room_for(bytes + 64, up);
                    HIP_OK(hipMemcpyAsync(f.text.p + f.text.used, b->text.p + text_lo, bytes, hipMemcpyHostToDevice, up));
                    h_start.reserve(nb); h_seq.reserve(nb); h_idl.reserve(nb); h_sql.reserve(nb); h_size.reserve(nb);
                    for (size_t k = 0; k < nb; ++k) {
                        h_start.p[k] = f.text.used + (r[k].start - text_lo); h_seq.p[k] = h_start.p[k] + r[k].id_len;
                        h_idl.p[k] = r[k].id_len; h_sql.p[k] = r[k].seq_len; h_size.p[k] = r[k].size;
                    }
                    f.start.room_for(nb, up); f.seq_off.room_for(nb, up); f.id_len.room_for(nb, up); f.seq_len.room_for(nb, up); f.size.room_for(nb, up);
                    HIP_OK(hipMemcpyAsync(f.start.p + f.n, h_start.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.seq_off.p + f.n, h_seq.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.id_len.p + f.n, h_idl.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.seq_len.p + f.n, h_sql.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.size.p + f.n, h_size.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipStreamSynchronize(up));            // the block and the staging arrays are reused
                    f.text.used += bytes;
                    f.start.used = f.seq_off.used = f.id_len.used = f.seq_len.used = f.size.used = f.n + nb;
                    f.n += nb;
                    side.pos += nb;
                }
                if (side.failed) parse_failure[s] = side.failure;
            } catch (...) { err[s] = std::current_exception(); }
        };
        CompressedOnDevice packed[2];
        bool on_device[2] = {false, false};
        bool plain_on_device[2] = {false, false};              // a plain regular file: copied to HBM as it is, cut into records there
        uint64_t plain_bytes[2] = {0, 0};
        auto fetch_or_load = [&](int s) {
            if (inflate_on_device()) {
                try {
                    if (has_gz_extension(in[s])) on_device[s] = fetch_bgzf(in[s], fetch_bytes, tuning_.device, packed[s], &dev[s]);
                    else plain_on_device[s] = fetch_plain(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                }
                catch (const DeviceOutOfMemory&) { err[s] = std::current_exception(); return; }   // rethrown below: the two-pass run takes over
                catch (const std::exception&) { on_device[s] = plain_on_device[s] = false; }          // the host way will say what is wrong
            }
            if (!on_device[s] && !plain_on_device[s]) { packed[s] = CompressedOnDevice(); dev[s].forget(); load(s); }
        };
        std::thread second(fetch_or_load, 1);
        fetch_or_load(0);
        second.join();
        for (int s = 0; s < 2; ++s) {
            if (!on_device[s] && !plain_on_device[s]) continue;
            StageClock::Scope t2("unordered/resident: inflate + record scan on the GPU");
            const bool ok = on_device[s] ? finish_on_device(eng.get(), stream, format_, packed[s], dev[s])
                                         : records_on_device(eng.get(), stream, format_, plain_bytes[s], dev[s]);
            if (!ok) {                                                                  // read it again the host way: that one reports
                dev[s].forget();
                packed[s] = CompressedOnDevice();
                load(s);
            }
        }
        for (int s = 0; s < 2; ++s) {
            if (err[s]) std::rethrow_exception(err[s]);
            if (parse_failure[s].set) { std::cerr << parse_failure[s].diag; throw std::runtime_error(parse_failure[s].what); }
        }
        for (int s = 0; s < 2; ++s) {
            FileOnDevice& f = dev[s];
            f.tag_off.reserve(f.n); f.tag_len.reserve(f.n);
            engine_ok(fqd_extract_tags(eng.get(), reinterpret_cast<const uint8_t*>(f.text.p), f.start.p, f.id_len.p, f.n, f.tag_off.p, f.tag_len.p));
        }
    }

    DeviceSide side[2];
    for (int s = 0; s < 2; ++s) {
        side[s].tag_bytes = side[s].seq_bytes = reinterpret_cast<const uint8_t*>(dev[s].text.p);
        side[s].tag_off = dev[s].tag_off.p; side[s].tag_len = dev[s].tag_len.p;
        side[s].seq_off = dev[s].seq_off.p; side[s].seq_len = dev[s].seq_len.p; side[s].n = dev[s].n;
    }
    // The join, the dedup and every buffer the writer needs come BEFORE the outputs exist: HBM that does not suffice for
    // them (DeviceOutOfMemory) still hands the job to the two-pass run.  On disk nothing differs from the reference's
    // order — outputs opened after the sort phase, then the merge (hpp:265-266) — a bad base found by the dedup cuts the
    // output at the same pair either way.
    fqd_engine* engine_now = eng.get();                        // (joins the helper thread: jp is ours from here on)
    join_and_dedup(engine_now, stream, side, tuning_.reference_tail_rule, jp);
    const uint64_t n_proc = jp.n_proc, upto = std::min<uint64_t>(n_proc, jp.written_below);
    uint64_t dups = 0;
    {
        std::vector<uint8_t> keep(upto);
        if (upto) HIP_OK(hipMemcpyAsync(keep.data(), jp.keep.p, upto, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (uint64_t k = 0; k < upto; ++k) dups += keep[k] == 0;
    }
    FileOnDevice* files[2] = {&dev[0], &dev[1]};
    const uint32_t* idx[2] = {jp.pair[0].p, jp.pair[1].p};
    SurvivorBuffers buffers;
    {
        const bool gz_out[2] = {has_gz_extension(out[0]), has_gz_extension(out[1])};
        plan_survivors(eng.get(), 2, files, idx, jp.keep.p, upto, gz_out, memlimit_, buffers);
    }

    OutputFile sink0(out[0]), sink1(out[1]);
    OutputFile* sinks[2] = {&sink0, &sink1};

    // ---- outputs: the device assembles windows of survivors in output order, the host writes them -----
    {
        StageClock::Scope t("unordered/resident: survivors out of HBM");
        write_survivors(eng.get(), stream, 2, files, idx, jp.keep.p, upto, dups, sinks, format_, memlimit_, true, &buffers);
    }
    if (tuning_.leave_memory_to_exit) g_leave_memory_to_exit = true;
    StageClock::report();
    if (jp.bad) throw_unknown_base(jp.bad_byte);
    summary_.total = n_proc; summary_.duplicates = dups; summary_.unmatched = jp.unmatched;
    if (verbose_) {
        std::cout << summary_.total << " valid read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        std::cout << summary_.unmatched << " Non-matching entries from both files were skipped.\n";
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
