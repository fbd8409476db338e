#include "records.hpp"

#include <algorithm>
#include <cstring>
#include <hip/hip_runtime_api.h>
#include <stdexcept>

namespace fqdhost {

namespace {

inline const char* find_nl(const char* b, const char* e)
{
    return b < e ? static_cast<const char*>(std::memchr(b, '\n', static_cast<size_t>(e - b))) : nullptr;
}

} // namespace

size_t scan_records(Format f, bool want_tag, const char* text, size_t n,
                    std::vector<RecordRef>& out, ParseFailure& fail)
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
        r.start = static_cast<uint64_t>(p - base);
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
        out.push_back(r);
        p += size;
    }
    return static_cast<size_t>(p - base);
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
    if (hipHostMalloc(&np, bytes, hipHostMallocDefault) != hipSuccess)
        throw std::runtime_error("cannot allocate pinned host memory for an input block");
    p = static_cast<char*>(np); cap = bytes;
}

RecordStream::RecordStream(const std::string& name, Format f, bool want_tag, size_t block_bytes)
    : file_(name), fmt_(f), want_tag_(want_tag), block_bytes_(block_bytes) {}

bool RecordStream::fill(Block& b)
{
    if (done_) return false;
    b.recs.clear(); b.used = 0; b.last = false; b.failure = ParseFailure(); b.held_back = false; b.first_record = n_records_;
    const size_t want = std::max(block_bytes_, carry_.size() + block_bytes_ / 2);
    b.text.reserve(want + 16);
    size_t have = carry_.size();
    if (have) std::memcpy(b.text.p, carry_.data(), have);
    carry_.clear();
    have += file_.read(b.text.p + have, want - have);
    size_t consumed = scan_records(fmt_, want_tag_, b.text.p, have, b.recs, b.failure);
    if (b.failure.set) {
        b.last = true; done_ = true;
        if (!b.recs.empty()) { b.recs.pop_back(); b.held_back = true; }      // fetched by the lookahead, never processed
    } else if (file_.eof()) {
        b.last = true; done_ = true;                                        // a trailing partial record is dropped
    } else {
        if (b.recs.size() < 2)                                              // one record larger than a whole block
            throw std::runtime_error("Not enough memory to read a single object!");
        consumed = b.recs.back().start;                                     // re-scan the last record with the next block
        b.recs.pop_back();
        carry_.assign(b.text.p + consumed, b.text.p + have);
    }
    b.used = consumed;
    n_records_ += b.recs.size();
    if (first_) {
        first_ = false;
        // bufferedinput.hpp:81-84: the very first record must parse.  A bad lead byte in
        // record 0 has already been reported through b.failure (read_new throws first).
        if (b.recs.empty() && !b.failure.set)
            throw std::runtime_error("Not enough memory to read a single object!");
    }
    return true;
}

} // namespace fqdhost
