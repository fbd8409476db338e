#include "records.hpp"
#include "file_io.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <cstring>
#include <emmintrin.h>
#include <hip/hip_runtime_api.h>
#include <stdexcept>
#include <thread>

namespace fqdhost {

namespace {

inline const char* find_nl(const char* b, const char* e)
{
    return b < e ? static_cast<const char*>(std::memchr(b, '\n', static_cast<size_t>(e - b))) : nullptr;
}

} // namespace

namespace {

// The scanner proper; emit(record) receives every complete, well-formed record in order.
template <class Emit>
size_t scan_core(Format f, bool want_tag, const char* text, size_t n, Emit&& emit, ParseFailure& fail, uint64_t start_base = 0)
{
    const char* const base = text;
    const char* const end = text + n;
    const char lead = f == Format::Fastq ? '@' : '>';
    const int n_lines = f == Format::Fastq ? 4 : 2;
    const char* p = text;
    while (p < end) {
        if (*p != lead) {                                  // fastqview.cpp:92,121-126 / fastaview.cpp:78,95-100
            fail.set = true;
            fail.diag = std::string("Invalid record start character: ") + *p + "\n";
            fail.what = f == Format::Fastq ? "Fastq record should start with @ symbol!"
                                           : "Fasta record should start with > symbol!";
            break;
        }
        const char* line = p;
        uint64_t len[4] = {0, 0, 0, 0};
        bool complete = true;
        for (int k = 0; k < n_lines; ++k) {                // fastqview.cpp:96-116: one '\n' search per line
            const char* nl = find_nl(line, end);
            if (!nl) { complete = false; break; }
            len[k] = static_cast<uint64_t>(nl - line) + 1;
            line = nl + 1;
        }
        if (!complete) break;                              // record continues in the next block
        const uint64_t size = len[0] + len[1] + len[2] + len[3];
        if (size > 0xFFFFFFFFull) throw std::runtime_error("Not enough memory to read a single object!");
        if (f == Format::Fastq && len[3] != len[1]) {      // fastqview.cpp:117,128-138 (lengths include '\n')
            fail.set = true;
            fail.diag = "Found sequence " + std::string(p + len[0], len[1] - 1) + " of length " + std::to_string(len[1]) +
                        " and quality string " + std::string(p + len[0] + len[1] + len[2], len[3] - 1) +
                        " of length " + std::to_string(len[3]) + "\n";
            fail.what = "Sequence and Quality fields of Fastq record should have the same length!";
            break;
        }
        RecordRef r;
        r.start = static_cast<uint64_t>(p - base) + start_base;
        r.size = static_cast<uint32_t>(size);
        r.id_len = static_cast<uint32_t>(len[0]);
        r.seq_len = static_cast<uint32_t>(len[1] - 1);
        r.tag_off = 0; r.tag_len = 0;
        if (want_tag) {                                    // fastqview.cpp:190-204
            const char* id_end = p + len[0];               // one past the ID line's '\n'
            const char* dot = static_cast<const char*>(std::memchr(p, '.', len[0]));
            const char* tag = dot ? dot + 1 : p + 1;
            const char* sp = tag < id_end ? static_cast<const char*>(std::memchr(tag, ' ', static_cast<size_t>(id_end - tag))) : nullptr;
            r.tag_off = static_cast<uint32_t>(tag - p);
            r.tag_len = static_cast<uint32_t>((sp ? sp : id_end) - tag);
        }
        emit(r);
        p += size;
    }
    return static_cast<size_t>(p - base);
}

} // namespace

size_t scan_records(Format f, bool want_tag, const char* text, size_t n,
                    std::vector<RecordRef>& out, ParseFailure& fail, uint64_t start_base)
{
    return scan_core(f, want_tag, text, n, [&](const RecordRef& r) { out.push_back(r); }, fail, start_base);
}

namespace {
double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::mutex g_clock_mutex;
std::map<std::string, std::pair<double, uint64_t>> g_clock;
} // namespace

bool StageClock::on() { static const bool v = std::getenv("FQD_HOST_TIMING") != nullptr; return v; }
void StageClock::add(const char* stage, double seconds)
{
    std::lock_guard<std::mutex> g(g_clock_mutex);
    auto& e = g_clock[stage]; e.first += seconds; ++e.second;
}
void StageClock::report()
{
    if (!on()) return;
    std::lock_guard<std::mutex> g(g_clock_mutex);
    std::fprintf(stderr, "[host timing] .gz members through %s\n", gz_codec_name());
    for (const auto& kv : g_clock) std::fprintf(stderr, "[host timing] %-28s %8.3f s  (%llu)\n", kv.first.c_str(), kv.second.first,
                                                static_cast<unsigned long long>(kv.second.second));
}
StageClock::Scope::Scope(const char* s) : stage(s), t0(on() ? now_seconds() : 0.0) {}
StageClock::Scope::~Scope() { if (on()) add(stage, now_seconds() - t0); }

unsigned host_threads()
{
    if (const char* v = std::getenv("FQD_HOST_THREADS")) return static_cast<unsigned>(std::max(1, std::atoi(v)));
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hw ? hw : 1u));      // measured on the GPU box (BGZF inflate + scan of one file): 8 -> 16 threads = 2.1 -> 1.5 s per 6.3 GB
}

namespace {

size_t count_newlines(const char* b, const char* e)
{
    size_t cnt = 0;
    const __m128i nl = _mm_set1_epi8('\n'), zero = _mm_setzero_si128();
    while (b + 16 <= e) {
        // matches are 0xFF = -1 per byte: subtract them into byte counters, 255 rounds at most
        __m128i acc = zero;
        const size_t rounds = std::min<size_t>(255, static_cast<size_t>(e - b) / 16);
        for (size_t r = 0; r < rounds; ++r, b += 16)
            acc = _mm_sub_epi8(acc, _mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b)), nl));
        const __m128i sums = _mm_sad_epu8(acc, zero);         // two 64-bit partial sums
        cnt += static_cast<size_t>(_mm_cvtsi128_si64(sums)) + static_cast<size_t>(_mm_cvtsi128_si64(_mm_srli_si128(sums, 8)));
    }
    for (; b < e; ++b) cnt += *b == '\n';
    return cnt;
}

} // namespace

size_t scan_records_parallel(Format f, bool want_tag, const char* text, size_t n,
                             std::vector<RecordRef>& out, ParseFailure& fail, unsigned threads, uint64_t start_base)
{
    static const size_t kMinSlice = [] {                   // FQD_SCAN_MIN_SLICE: test hook
        const char* v = std::getenv("FQD_SCAN_MIN_SLICE");
        return v ? static_cast<size_t>(std::max(1L, std::atol(v))) : size_t(4u << 20);
    }();
    const unsigned parts = static_cast<unsigned>(std::min<size_t>(threads, n / kMinSlice));
    if (parts <= 1) return scan_records(f, want_tag, text, n, out, fail, start_base);
    const size_t n_lines = f == Format::Fastq ? 4 : 2;

    // 1. newlines per slice -> index of the line each slice starts in
    std::vector<size_t> cut(parts + 1), lines(parts);
    for (unsigned p = 0; p <= parts; ++p) cut[p] = n / parts * p;
    cut[parts] = n;
    run_parts(parts, [&](unsigned p) { lines[p] = count_newlines(text + cut[p], text + cut[p + 1]); });

    // 2. first record start at or after each cut (a line start whose index is a multiple of
    //    n_lines) and how many records start before it
    std::vector<size_t> start(parts + 1, n), first_rec(parts + 1, 0);
    size_t before = 0;                                     // newlines in text[0, cut[p])
    for (unsigned p = 0; p < parts; ++p) {
        const char* q = text + cut[p];
        size_t line = before;                              // index of the line q lies in
        if (cut[p] != 0 && q[-1] != '\n') {                // inside a line: go to the next line start
            const char* nl = find_nl(q, text + n);
            q = nl ? nl + 1 : nullptr; ++line;
        }
        while (q && line % n_lines != 0) {
            const char* nl = find_nl(q, text + n);
            q = nl ? nl + 1 : nullptr; ++line;
        }
        start[p] = q ? static_cast<size_t>(q - text) : n;
        first_rec[p] = line / n_lines;
        before += lines[p];
    }
    first_rec[parts] = before / n_lines;                   // complete records in the whole text, if all are well formed
    for (unsigned p = parts; p-- > 0;)                     // a slice that lies inside one long record owns nothing
        if (start[p] >= start[p + 1]) { start[p] = start[p + 1]; first_rec[p] = first_rec[p + 1]; }

    // 3. every slice writes its records straight to their final places
    const size_t base = out.size();
    out.resize(base + first_rec[parts]);
    std::vector<size_t> found(parts, 0), consumed(parts, 0);
    std::vector<ParseFailure> failed(parts);
    run_parts(parts, [&](unsigned p) {
        RecordRef* dst = out.data() + base + first_rec[p];
        const size_t room = first_rec[p + 1] - first_rec[p], off = start[p];
        size_t k = 0;
        consumed[p] = scan_core(f, want_tag, text + off, start[p + 1] - off,
                                [&](const RecordRef& r) { if (k < room) dst[k] = r; ++k; }, failed[p], start_base + off);
        found[p] = k;
    });
    for (unsigned p = 0; p < parts; ++p) {
        const bool whole = !failed[p].set && consumed[p] == start[p + 1] - start[p] && found[p] == first_rec[p + 1] - first_rec[p];
        if (whole) continue;
        // a malformed record, or the incomplete record at the end of the text: the scan ends here
        out.resize(base + first_rec[p] + std::min(found[p], first_rec[p + 1] - first_rec[p]));
        if (failed[p].set) fail = failed[p];
        return start[p] + consumed[p];
    }
    return start[parts];
}

int compare_tags(const char* a, uint32_t alen, const char* b, uint32_t blen)
{
    const int c = std::strncmp(a, b, std::min(alen, blen));
    if (c == 0 && alen != blen) return alen < blen ? -1 : 1;
    return c;
}

PinnedBuffer::~PinnedBuffer() { if (p) (void)hipHostFree(p); }

void PinnedBuffer::reserve(size_t bytes)
{
    if (bytes <= cap) return;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    void* np = nullptr;
    if (hipHostMalloc(&np, bytes, hipHostMallocPortable) != hipSuccess)
        throw std::runtime_error("cannot allocate pinned host memory for an input block");
    p = static_cast<char*>(np); cap = bytes;
}

RecordStream::RecordStream(const std::string& name, Format f, bool want_tag, size_t block_bytes)
    : file_(name), fmt_(f), want_tag_(want_tag), block_bytes_(block_bytes) {}

// Stage 1: the next raw bytes of the file, behind kHeadroom bytes left free for what the block
// before it carries over.  Touches only the file, so it can run ahead of stage 2.
bool RecordStream::read_raw(Block& b)
{
    if (raw_done_) return false;
    b.recs.clear(); b.used = 0; b.last = false; b.failure = ParseFailure(); b.held_back = false;
    { StageClock::Scope t("reader: pinned alloc"); b.text.reserve(kHeadroom + block_bytes_ + 16); }
    { StageClock::Scope t("reader: file read"); b.raw_len = file_.read(b.text.p + kHeadroom, block_bytes_, host_threads()); }
    b.raw_eof = file_.eof();
    if (b.raw_eof) raw_done_ = true;
    return true;
}

// Stage 2, block after block in file order: the carried-over tail of the previous block goes in
// front of the raw bytes, the whole is scanned, the incomplete tail is carried on.
void RecordStream::finish(Block& b)
{
    b.first_record = n_records_;
    if (done_) { b.last = true; return; }                                   // a block read ahead of a failure: nothing in it counts
    size_t carry = carry_.size();
    if (carry > kHeadroom) {                                                // a record of more than a MiB: make room the slow way
        PinnedBuffer bigger;
        bigger.reserve(carry + b.raw_len + 16);
        std::memcpy(bigger.p + carry, b.text.p + kHeadroom, b.raw_len);
        std::swap(bigger.p, b.text.p); std::swap(bigger.cap, b.text.cap);
        std::memcpy(b.text.p, carry_.data(), carry);
    } else if (carry) {
        std::memcpy(b.text.p + kHeadroom - carry, carry_.data(), carry);
    }
    const size_t off = carry > kHeadroom ? 0 : kHeadroom - carry;
    const char* text = b.text.p + off;
    const size_t have = carry + b.raw_len;
    carry_.clear();
    size_t consumed;
    { StageClock::Scope t("reader: record scan"); consumed = scan_records_parallel(fmt_, want_tag_, text, have, b.recs, b.failure, host_threads(), off); }
    if (b.failure.set) {
        b.last = true; done_ = true;
        if (!b.recs.empty()) { b.recs.pop_back(); b.held_back = true; }      // fetched by the lookahead, never processed
    } else if (b.raw_eof) {
        b.last = true; done_ = true;                                        // a trailing partial record is dropped
    } else {
        if (b.recs.size() < 2)                                              // one record larger than a whole block
            throw std::runtime_error("Not enough memory to read a single object!");
        consumed = b.recs.back().start - off;                               // re-scan the last record with the next block
        b.recs.pop_back();
        carry_.assign(text + consumed, text + have);
    }
    b.used = off + consumed;
    n_records_ += b.recs.size();
    if (first_) {
        first_ = false;
        // bufferedinput.hpp:81-84: the very first record must parse.  A bad lead byte in
        // record 0 has already been reported through b.failure (read_new throws first).
        if (b.recs.empty() && !b.failure.set)
            throw std::runtime_error("Not enough memory to read a single object!");
    }
}

bool RecordStream::fill(Block& b)
{
    if (done_ || !read_raw(b)) return false;
    finish(b);
    return true;
}

} // namespace fqdhost
