// file_io.hpp — plain / gzip input and output files for the host driver.
// Mirrors FileUtils::I_InputFile{TXT,GZ}, openInputFile and UniversalOutputFile of the
// reference (src/file_utils.hpp:25-79, src/file_utils.cpp:53-92): a ".gz" extension selects
// gzip, anything else is plain; same diagnostics when a file cannot be opened.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <future>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

namespace fqdhost {

bool has_gz_extension(const std::string& name);               // file_utils.cpp:42-48

// An error that comes with a line for stderr ahead of the exception text, the way the reference
// prints a diagnostic and then throws.  Whoever lets it reach main() prints `diag` first, so
// work done on helper threads never prints out of order.
struct DiagnosedError : std::runtime_error {
    std::string diag;
    DiagnosedError(std::string d, const std::string& what) : std::runtime_error(what), diag(std::move(d)) {}
};

// "Cannot open file <name>" + "File does not exist or cannot be opened!", as check_fstream_ok
// (file_utils.hpp:111-121).
[[noreturn]] void throw_cannot_open(const std::string& name);

// Length of the BGZF member that starts at p (at least 18 bytes readable), or 0 if p does not start one: a gzip
// header with an extra field holding the 'B','C' subfield = total size - 1.  *data_off = where its deflate data starts.
size_t bgzf_member_size(const unsigned char* p, size_t avail, size_t* data_off);

// "libdeflate" when the system library was found (and FQD_CODEC != zlib), else "zlib": what BGZF members go through.
const char* gz_codec_name();

class InputFile {
public:
    // as_bytes: the file as it lies on disk, whatever its extension (for callers that inflate it themselves).
    explicit InputFile(const std::string& name, bool as_bytes = false);
    ~InputFile();
    InputFile(const InputFile&) = delete;
    InputFile& operator=(const InputFile&) = delete;
    // Reads up to n bytes; fewer only at end of file (eof() turns true).  Large reads of a
    // regular plain file are split over `threads` preads so the copy out of the page cache is
    // not bound by one core.
    size_t read(char* dst, size_t n, unsigned threads = 1);
    bool eof() const { return eof_; }
private:
    size_t read_bgzf(char* dst, size_t n, unsigned threads);
    bool fill_compressed();
    bool gz_; gzFile g_ = nullptr; int fd_ = -1; bool eof_ = false;
    bool undecided_ = false;                                    // an ordinary .gz whose reader (several threads, or zlib) the first read picks
    void* pgzip_ = nullptr;                                     // pgz::Reader (pgzip.hpp), when it was the pick
    bool regular_ = false; uint64_t size_ = 0, offset_ = 0;     // plain regular files: known size, own cursor
    // BGZF (blocked gzip, what bgzip / sequencer software write): every member says how long it
    // is, so a batch of members is cut out of the compressed stream without inflating and then
    // inflated on several threads, each member straight to its place in the caller's buffer.
    bool bgzf_ = false, comp_eof_ = false;
    std::vector<unsigned char> comp_;                           // compressed bytes not yet consumed
    size_t comp_pos_ = 0;
    std::vector<char> spill_;                                   // inflated tail of a member that did not fit the last read
    size_t spill_pos_ = 0;
};

// Plain files go through a 256 KiB buffer (file_utils.cpp:90).  ".gz" files are written as BGZF:
// independent gzip members of at most 65280 input bytes, each carrying its compressed size in the
// header's extra field, closed by the empty end-of-file member.  Batches of members are deflated
// on worker threads and written in order: the decompressed content is what the reference would
// have written (file_utils.cpp:87-88; gzip readers concatenate members), any gzip tool reads it,
// and BGZF-aware readers (this one included) can inflate it in parallel.
class OutputFile {
public:
    explicit OutputFile(const std::string& name);
    ~OutputFile();
    OutputFile(const OutputFile&) = delete;
    OutputFile& operator=(const OutputFile&) = delete;
    void write(const char* p, size_t n);
    // The same for a large buffer lent until the call returns: .gz members are deflated straight out of it.
    void write_borrowed(const char* p, size_t n);
    // .gz only: whole gzip members made elsewhere (on the GPU: fqd_bgzf_deflate) go to the file as they are,
    // after whatever write() still holds.
    void write_members(const char* p, size_t n, unsigned threads = 1);
    bool is_gz() const { return gz_; }
    // Several pieces in one go: plain files hand them to writev() as they lie (no staging copy).
    struct Piece { const char* p; size_t n; };
    // threads > 1: a large batch into a regular plain file is written by that many threads at once (pwritev).
    void write_pieces(const Piece* pieces, size_t count, unsigned threads = 1);
    void close();
private:
    void submit_block();
    void drain(size_t keep_in_flight);
    void flush_plain();
    void put_plain(const char* p, size_t n);
    bool gz_; FILE* f_ = nullptr; std::string name_;
    int fd_ = -1;                                         // plain: raw descriptor + own 256 KiB buffer
    bool regular_ = false;                                // plain: a regular file (seekable: parallel pwritev allowed)
    std::string plain_buf_;
    std::string block_;                                   // gz: bytes of the member being filled
    std::deque<std::future<std::string>> in_flight_;      // gz: members being deflated, oldest first
    size_t max_in_flight_ = 8;
};

} // namespace fqdhost
