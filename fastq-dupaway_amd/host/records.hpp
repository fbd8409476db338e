// records.hpp — the host feeder: block reader + FASTQ/FASTA record scanner.
// Mirrors BufferedInput<T> (src/bufferedinput.hpp:8-103) and the record views
// FastqView / FastaView (+WithId) of the reference (src/fastqview.cpp:89-138,190-204,
// src/fastaview.cpp:75-100,153-167): same record grammar, same validation errors and
// diagnostics, same ID-tag rule.  Unlike the reference it scans a whole block at once
// and keeps the raw bytes in pinned memory, so the block can be DMA'd to HBM as is.
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "file_io.hpp"

namespace fqdhost {

enum class Format { Fastq, Fasta };

// One record inside a block; positions are byte offsets into the block's text.
struct RecordRef {
    RecordRef() {}       // left uninitialised on purpose: vectors of millions are resized, then filled by several threads
    uint64_t start;      // first byte of the ID line ('@' or '>')
    uint32_t size;       // whole record, every newline included (what survivors are written with)
    uint32_t id_len;     // ID line including its '\n'
    uint32_t seq_len;    // bases, WITHOUT the '\n' (the reference passes seq_len()-1 to its keys)
    uint32_t tag_off;    // join key (--unordered): offset from `start`, length — fastqview.cpp:190-204
    uint32_t tag_len;
    uint64_t seq_start() const { return start + id_len; }
};

// A malformed record: the stderr line the reference prints and its exception text.
struct ParseFailure {
    bool        set = false;
    std::string diag;    // e.g. "Invalid record start character: x\n"
    std::string what;    // e.g. "Fastq record should start with @ symbol!"
};

// Scans complete records in text[0,n).  Appends to `out`, returns the bytes consumed
// (start of the first incomplete record).  Stops at a malformed record and fills `fail`.
// start_base is added to every RecordRef::start (text lies that far into its block).
size_t scan_records(Format f, bool want_tag, const char* text, size_t n,
                    std::vector<RecordRef>& out, ParseFailure& fail, uint64_t start_base = 0);

// The same scan on several threads, with the same result: lines are counted per slice first, so
// every slice knows which of its line starts begin a record (every 4th / 2nd line of the file),
// then the slices are scanned independently and stitched together in order; the earliest
// malformed record wins, exactly as when scanning from the front.  Falls back to scan_records
// for small inputs or threads <= 1.
size_t scan_records_parallel(Format f, bool want_tag, const char* text, size_t n,
                             std::vector<RecordRef>& out, ParseFailure& fail, unsigned threads, uint64_t start_base = 0);

// FQD_HOST_TIMING=1: wall time per host stage, summed over the run and printed to stderr at exit.
struct StageClock {
    static bool on();
    static void add(const char* stage, double seconds);
    static void report();
    struct Scope {
        const char* stage; double t0;
        explicit Scope(const char* s);
        ~Scope();
    };
};

// Runs body(0..parts-1), part 0 on the calling thread; an exception of any part is rethrown.
template <class Body>
void run_parts(unsigned parts, Body&& body)
{
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> err(parts);
    auto guarded = [&](unsigned p) { try { body(p); } catch (...) { err[p] = std::current_exception(); } };
    for (unsigned p = 1; p < parts; ++p) pool.emplace_back(guarded, p);
    guarded(0);
    for (std::thread& t : pool) t.join();
    for (unsigned p = 0; p < parts; ++p) if (err[p]) std::rethrow_exception(err[p]);
}

// Worker threads a host stage may use: FQD_HOST_THREADS, else min(16, hardware threads).
unsigned host_threads();

// Order of two ID tags: strncmp over the shorter, then shorter first (fastqview.cpp:168-178).
int compare_tags(const char* a, uint32_t alen, const char* b, uint32_t blen);

// Page-locked host memory (hipHostMalloc) so blocks can be copied to HBM asynchronously.
struct PinnedBuffer {
    char*  p = nullptr;
    size_t cap = 0;
    PinnedBuffer() = default;
    ~PinnedBuffer();
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    void reserve(size_t bytes);          // contents are NOT preserved
};

// One block of input: raw text of complete records + their index.
struct Block {
    PinnedBuffer           text;
    size_t                 used = 0;       // end of the complete records in text (they start at recs[0].start, not at 0)
    size_t                 raw_len = 0;    // RecordStream stage 1: raw bytes read behind the headroom, and whether the file ended
    bool                   raw_eof = false;
    std::vector<RecordRef> recs;
    uint64_t               first_record = 0;   // index of recs[0] within the file
    bool                   last = false;       // nothing follows (end of file or failure)
    // failure.set: a malformed record ends the stream.  The reference parses one record
    // ahead (BufferedInput::next, bufferedinput.hpp:91-103), so the good record just before
    // the malformed one is fetched but never processed: it is withheld from recs
    // (held_back) unless the malformed record is the very first of the file.
    ParseFailure           failure;
    bool                   held_back = false;
};

// Streams a file as blocks.  Every record handed out is known to be followed by a
// well-formed record or by the end of the file (the last record of a non-final block
// is carried into the next block), so a failure never has to be applied retroactively.
// The first block must hold at least one complete record,
// else "Not enough memory to read a single object!" (bufferedinput.hpp:82-84), which is
// also what an empty file gives.  A final record without '\n' is dropped silently
// (reference README.md:178).
class RecordStream {
public:
    RecordStream(const std::string& name, Format f, bool want_tag, size_t block_bytes);
    // Fills b (reusing its memory); returns false when the stream had already ended.
    bool fill(Block& b);
    // The same in two stages, so that a reader thread can fetch block k+1 from the file while another
    // scans block k: read_raw (file only; false when the file had already ended) then finish (in file
    // order: joins the block to its predecessor's carried-over tail and scans it).
    static constexpr size_t kHeadroom = 1u << 20;
    bool read_raw(Block& b);
    void finish(Block& b);
    uint64_t records_so_far() const { return n_records_; }
private:
    InputFile         file_;
    Format            fmt_;
    bool              want_tag_;
    size_t            block_bytes_;
    std::vector<char> carry_;              // incomplete record left over from the previous block
    uint64_t          n_records_ = 0;
    bool              first_ = true, done_ = false, raw_done_ = false;
};

} // namespace fqdhost
