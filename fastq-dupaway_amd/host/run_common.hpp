// run_common.hpp — what the ways of running the `--fast` path share (host/run_*.cpp, survivor_writer.cpp): typed device
// errors, RAII over HIP memory and the C ABI, the block pipeline of an input file (Side), a batch on its way from the
// feeder over the GPU to the writers (Work), the writer threads, and the GPU-resident runs' view of a file.
// Everything lives in fqdhost::detail; the public surface stays hash_dup_remover.hpp.
#pragma once
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
namespace detail {


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
inline void announce_handover(const char* run, const std::exception& e)
{
    std::cerr << "[fastq-dupaway] " << run << " gave up on a GPU error (" << e.what() << "); continuing with the streaming run\n";
}

// RAII over the C ABI
// Set by a resident run of the CLI after its outputs are closed (Tuning::leave_memory_to_exit): from then on
// buffers are not freed one by one — the process is about to end and the driver releases everything at once.
inline std::atomic<bool> g_leave_memory_to_exit{false};

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
[[noreturn]] inline void throw_unknown_base(uint8_t byte)
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

// Writer threads (one per output file: gzip outputs deflate in parallel): survivors, verbatim, in
// the order the batches are handed over (survivor_writer.cpp).
unsigned write_threads();

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
    void body(int s);
    int S_; std::unique_ptr<OutputFile>* sink_; Channel<Work>* recycle_;
    Channel<Work> queue_[2]; std::thread thread_[2]; std::exception_ptr error_[2];
    Work last_resort_; Channel<Work> nowhere_;
};

// ---- `--unordered` (run_unordered.cpp) ----------------------------------------------------------------------------

// Copies entry k of a device array of uint32.
uint32_t peek_u32(const uint32_t* d, uint64_t k, hipStream_t s);

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

void join_and_dedup(fqd_engine* e, hipStream_t stream, const DeviceSide (&side)[2], bool tail_rule, JoinedPairs& jp);

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

bool is_regular_file(const std::string& name, uint64_t& size);

// ---- text resident in HBM (run_resident.cpp, survivor_writer.cpp) -------------------------------------------------

// One pass, text resident in HBM (see run_unordered).
// A file whose text stays in HBM: the text and, per record, where it starts, where its sequence starts, the
// lengths of its ID line and sequence, its size.
struct FileOnDevice {
    GrowDevice<char> text;
    GrowDevice<uint64_t> start, seq_off; GrowDevice<uint32_t> id_len, seq_len, size;
    Device<uint64_t> tag_off; Device<uint32_t> tag_len;
    // what inflated an ordinary gzip file here (fetch_gzip_ordinary): the packed bytes and the engine whose scratch the inflating
    // used stay until the run ends — gigabytes freed in mid-run are cleared by the driver while the process's next hipMalloc waits
    Device<char> packed; std::shared_ptr<EngineHandle> codec;
    uint64_t n = 0;
    void forget() { text.used = start.used = seq_off.used = id_len.used = seq_len.used = size.used = 0; n = 0; }
    void release() { packed.release(); codec.reset(); text.release(); start.release(); seq_off.release(); id_len.release(); seq_len.release(); size.release(); tag_off.release(); tag_len.release(); n = 0; }
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

bool inflate_on_device();                                     // FQD_GUNZIP_DEVICE
bool deflate_on_device();                                     // FQD_GZ_DEVICE / FQD_GZ_LEVEL
bool fetch_bgzf(const std::string& name, size_t block_bytes, int device, CompressedOnDevice& c, FileOnDevice* into = nullptr);
bool fetch_plain(const std::string& name, size_t block_bytes, int device, FileOnDevice& f, uint64_t& text_bytes);
bool fetch_gzip_ordinary(const std::string& name, size_t block_bytes, int device, FileOnDevice& f, uint64_t& text_bytes);   // FQD_GUNZIP_ORDINARY_DEVICE=0: off
bool records_on_device(fqd_engine* e, hipStream_t stream, Format format, uint64_t text_bytes, FileOnDevice& f);
bool finish_on_device(fqd_engine* e, hipStream_t stream, Format format, CompressedOnDevice& c, FileOnDevice& f);
void guess_capacity(int S, const std::string* in, uint64_t& reads, uint64_t& bases);

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

void plan_survivors(fqd_engine* e, int S, FileOnDevice* const* file, const uint32_t* const* idx, const uint8_t* keep, uint64_t upto,
                    const bool* gz_out, long long memlimit, SurvivorBuffers& b);
void write_survivors(fqd_engine* e, hipStream_t stream, int S, FileOnDevice* const* file, const uint32_t* const* idx,
                     const uint8_t* keep, uint64_t upto, uint64_t dups, OutputFile* const* sinks, Format format, long long memlimit,
                     bool close_sinks = true, SurvivorBuffers* planned = nullptr);

} // namespace detail
} // namespace fqdhost
