// fqd_oracle.cpp — see fqd_oracle.hpp.  TEST INFRASTRUCTURE ONLY.
#include "fqd_oracle.hpp"

#include <algorithm>
#include <cstring>
#include <iostream>
#include <memory>
#include <unordered_set>
#include <zlib.h>

namespace fqo {

// ======================= a1-a3: packing ===================================

// seq_utils.cpp:3-21 — A0 C1 G2 T3 N4; anything else is reported on stderr and thrown.
int base_digit(char c)
{
    if (c == 'A') return 0;
    if (c == 'C') return 1;
    if (c == 'G') return 2;
    if (c == 'T') return 3;
    if (c == 'N') return 4;
    std::cerr << "Error: unknown character in DNA sequence: " << c << '\n';
    throw UnknownBase(c);
}

// seq_utils.cpp:23-33 — Horner evaluation in base 5, most significant digit first.
uint64_t pack_chunk(const char* s, size_t n)
{
    uint64_t v = 0;
    for (const char* p = s; p != s + n; ++p)
        v = v * 5u + static_cast<uint64_t>(base_digit(*p));
    return v;
}

// seq_utils.cpp:35-49 — ceil(len/17) chunks, the last one may be short.
void pack_sequence(std::vector<uint64_t>& out, const char* s, ssize_t len)
{
    const long n_chunks = (len + kChunkBases - 1) / kChunkBases;
    out.reserve(static_cast<size_t>(n_chunks));
    for (ssize_t pos = 0; pos < len; pos += kChunkBases)
        out.push_back(pack_chunk(s + pos, static_cast<size_t>(std::min<ssize_t>(kChunkBases, len - pos))));
}

// ======================= a4-a7: keys ======================================

SingleKey::SingleKey(const char* s, ssize_t n) : len(n) { pack_sequence(words, s, n); }

bool SingleKey::operator==(const SingleKey& o) const
{
    return len == o.len && words == o.words;
}

PairKey::PairKey(const char* l, ssize_t ln, const char* r, ssize_t rn) : llen(ln), rlen(rn)
{
    pack_sequence(lwords, l, ln);
    pack_sequence(rwords, r, rn);
}

bool PairKey::operator==(const PairKey& o) const
{
    return llen == o.llen && rlen == o.rlen && lwords == o.lwords && rwords == o.rwords;
}

// boost::hash_combine as used at hash_dup_remover.hpp:49,60,62,64 (Boost 1.81.0,
// pinned by the reference's Dockerfile:17-20).  Boost is not in this image, so
// this is the published 64-bit form restated from memory:
//     seed = mix(seed + 0x9e3779b9 + v),  mix = xorshift32 / *m / xorshift32 / *m / xorshift28.
// It only picks buckets: survivors do not depend on it.
static inline void bucket_mix(size_t& seed, uint64_t v)
{
    uint64_t x = seed + 0x9e3779b9ull + v;
    const uint64_t m = (uint64_t(0xe9846af) << 32) + 0x9b1a615dull;
    x ^= x >> 32; x *= m;
    x ^= x >> 32; x *= m;
    x ^= x >> 28;
    seed = x;
}

size_t SingleKeyHash::operator()(const SingleKey& k) const
{
    size_t seed = k.words.size();
    for (uint64_t w : k.words) bucket_mix(seed, w);
    return seed;
}

size_t PairKeyHash::operator()(const PairKey& k) const
{
    size_t seed = k.lwords.size();
    for (uint64_t w : k.lwords) bucket_mix(seed, w);
    bucket_mix(seed, k.rwords.size());
    for (uint64_t w : k.rwords) bucket_mix(seed, w);
    return seed;
}

using SingleSet = std::unordered_set<SingleKey, SingleKeyHash>;   // hash_dup_remover.hpp:70
using PairSet   = std::unordered_set<PairKey, PairKeyHash>;       // hash_dup_remover.hpp:71
constexpr size_t kReserve = 1000u * 1000u;                       // hash_dup_remover.hpp:17,114

// ======================= a13/a14: records ==================================

static const char* line_end(const char* b, const char* e)
{
    const void* p = (b < e) ? std::memchr(b, '\n', static_cast<size_t>(e - b)) : nullptr;
    return p ? static_cast<const char*>(p) : e;
}

ssize_t parse_record(Format f, bool want_tag, const char* b, const char* e, Record& r)
{
    r = Record();
    if (b >= e) return -1;                                   // fastqview.cpp:91 / fastaview.cpp:77
    const char lead = (f == FASTQ) ? '@' : '>';
    if (*b != lead) {                                        // fastqview.cpp:92,121-126 / fastaview.cpp:78,95-100
        std::cerr << "Invalid record start character: " << *b << std::endl;
        throw std::runtime_error(f == FASTQ ? "Fastq record should start with @ symbol!"
                                            : "Fasta record should start with > symbol!");
    }
    const int n_lines = (f == FASTQ) ? 4 : 2;
    ssize_t lens[4] = {0, 0, 0, 0};
    const char* p = b;
    for (int i = 0; i < n_lines; ++i) {                      // 4 (2) newline searches: fastqview.cpp:96-116
        const char* nl = line_end(p, e);
        if (nl == e) return -1;
        lens[i] = nl - p + 1;
        p = nl + 1;
    }
    r.at = b; r.id_len = lens[0]; r.seq_len = lens[1]; r.plus_len = lens[2]; r.qual_len = lens[3];
    if (f == FASTQ && r.qual_len != r.seq_len) {             // fastqview.cpp:117,128-138
        std::cerr << "Found sequence ";
        std::cerr.write(r.seq(), r.seq_len - 1);
        std::cerr << " of length " << r.seq_len << " and quality string ";
        std::cerr.write(r.seq() + r.seq_len + r.plus_len, r.qual_len - 1);
        std::cerr << " of length " << r.qual_len << std::endl;
        throw std::runtime_error("Sequence and Quality fields of Fastq record should have the same length!");
    }
    if (want_tag) {                                          // fastqview.cpp:190-204 / fastaview.cpp:153-167
        const char* id_end = b + r.id_len;                   // one past the ID line's '\n'
        const char* dot = static_cast<const char*>(std::memchr(b, '.', static_cast<size_t>(r.id_len)));
        r.tag = dot ? dot + 1 : b + 1;
        const char* sp = (r.tag < id_end)
            ? static_cast<const char*>(std::memchr(r.tag, ' ', static_cast<size_t>(id_end - r.tag))) : nullptr;
        r.tag_len = (sp ? sp : id_end) - r.tag;
    }
    return r.bytes();
}

// fastqview.cpp:168-178 — strncmp over the shorter length, then shorter first.
static int compare_tag_bytes(const char* a, ssize_t alen, const char* b, ssize_t blen)
{
    int c = std::strncmp(a, b, static_cast<size_t>(std::min(alen, blen)));
    if (c == 0 && alen != blen) return alen < blen ? -1 : 1;
    return c;
}
int compare_tags(const Record& a, const Record& b)
{
    return compare_tag_bytes(a.tag, a.tag_len, b.tag, b.tag_len);
}

// ======================= a16: files =========================================

static bool has_gz_ext(const std::string& name)              // file_utils.cpp:42-48
{
    // std::filesystem::path::extension(): text from the last '.' of the filename.
    size_t slash = name.find_last_of('/');
    std::string base = (slash == std::string::npos) ? name : name.substr(slash + 1);
    size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return false;
    return base.substr(dot) == ".gz";
}

[[noreturn]] static void cannot_open(const std::string& name) // file_utils.hpp:111-121
{
    std::cerr << "Cannot open file " << name << std::endl;
    throw std::runtime_error("File does not exist or cannot be opened!");
}

class InFile {                                               // file_utils.cpp:53-79
public:
    explicit InFile(const std::string& name) : gz_(has_gz_ext(name))
    {
        if (gz_) { g_ = gzopen(name.c_str(), "rb"); if (!g_) cannot_open(name); gzbuffer(g_, 1 << 20); }
        else     { f_ = std::fopen(name.c_str(), "rb"); if (!f_) cannot_open(name); }
    }
    ~InFile() { if (g_) gzclose(g_); if (f_) std::fclose(f_); }
    InFile(const InFile&) = delete;
    // Reads up to n bytes; sets eof() once fewer than n could be delivered.
    ssize_t read(char* dst, ssize_t n)
    {
        ssize_t got = 0;
        while (got < n) {
            ssize_t k;
            if (gz_) {
                int want = static_cast<int>(std::min<ssize_t>(n - got, 1 << 30));
                k = gzread(g_, dst + got, static_cast<unsigned>(want));
                if (k < 0) throw std::runtime_error("gzip stream is corrupt");
            } else {
                k = static_cast<ssize_t>(std::fread(dst + got, 1, static_cast<size_t>(n - got), f_));
            }
            if (k <= 0) { eof_ = true; break; }
            got += k;
        }
        return got;
    }
    bool eof() const { return eof_; }
private:
    bool gz_; gzFile g_ = nullptr; FILE* f_ = nullptr; bool eof_ = false;
};

class OutFile {                                              // file_utils.cpp:83-92
public:
    explicit OutFile(const std::string& name) : gz_(has_gz_ext(name))
    {
        if (gz_) { g_ = gzopen(name.c_str(), "wb"); if (!g_) cannot_open(name); gzbuffer(g_, 64 * 1024); }
        else     { f_ = std::fopen(name.c_str(), "wb"); if (!f_) cannot_open(name); std::setvbuf(f_, nullptr, _IOFBF, 256 * 1024); }
    }
    ~OutFile() { if (g_) gzclose(g_); if (f_) std::fclose(f_); }
    OutFile(const OutFile&) = delete;
    void write(const char* p, ssize_t n)
    {
        if (n <= 0) return;
        if (gz_) gzwrite(g_, p, static_cast<unsigned>(n));
        else     std::fwrite(p, 1, static_cast<size_t>(n), f_);
    }
private:
    bool gz_; gzFile g_ = nullptr; FILE* f_ = nullptr;
};

// ======================= a15: block reader ==================================

// bufferedinput.hpp:8-103.  A fixed block plus ONE record of lookahead: next()
// hands out the record parsed earlier and parses the following one, so a
// malformed record k+1 aborts the run before record k is processed, exactly
// as in the reference.  Differences, both outside the reference's correct
// envelope: a refresh never discards unread records, and a file
// whose size is an exact multiple of the block does not throw.
class BlockReader {
public:
    BlockReader(const std::string& name, Format f, bool want_tag, ssize_t block)
        : file_(name), fmt_(f), want_tag_(want_tag), cap_(block), buf_(new char[static_cast<size_t>(block)])
    {
        size_ = file_.read(buf_.get(), cap_);
        pos_ = 0;
        ssize_t n = parse_record(fmt_, want_tag_, buf_.get(), buf_.get() + size_, ahead_);
        if (n < 0) throw std::runtime_error("Not enough memory to read a single object!"); // bufferedinput.hpp:82-84
        pos_ = n;
    }
    bool done() const { return ahead_.empty(); }             // nothing left to hand out
    // True when the record just handed out was the last one parseable in this
    // block (bufferedinput.hpp:96-99).
    bool block_end() const { return block_end_; }
    // Returns the lookahead record and parses the next; the returned view stays
    // valid until refill().
    Record next()
    {
        Record cur = ahead_;
        ssize_t n = parse_record(fmt_, want_tag_, buf_.get() + pos_, buf_.get() + size_, ahead_);
        if (n < 0) { ahead_ = Record(); block_end_ = true; }
        else       { pos_ += n; }
        return cur;
    }
    // bufferedinput.hpp:58-88 — slide the unread tail to the front and top up.
    // Call only when block_end() (every handed-out record has been consumed).
    void refill()
    {
        if (file_.eof()) return;                             // stays at block_end; done() is true
        ssize_t tail = size_ - pos_;
        std::memmove(buf_.get(), buf_.get() + pos_, static_cast<size_t>(tail));
        size_ = tail + file_.read(buf_.get() + tail, cap_ - tail);
        pos_ = 0;
        block_end_ = false;
        ssize_t n = parse_record(fmt_, want_tag_, buf_.get(), buf_.get() + size_, ahead_);
        if (n < 0) {
            if (file_.eof()) { ahead_ = Record(); block_end_ = true; return; }  // trailing partial record: dropped
            throw std::runtime_error("Not enough memory to read a single object!");
        }
        pos_ = n;
    }
private:
    InFile file_; Format fmt_; bool want_tag_; ssize_t cap_;
    std::unique_ptr<char[]> buf_;
    ssize_t size_ = 0, pos_ = 0;
    Record ahead_; bool block_end_ = false;
};

// ======================= a9: single-end driver ==============================

// hash_dup_remover.hpp:105-148.  find-then-insert on an unordered_set of
// (len, base-5 words); the first occurrence is written verbatim, in input order.
Summary filter_single(const std::string& in, const std::string& out, const Options& o)
{
    OutFile sink(out);                                       // hpp:110 (created before the input is opened)
    SingleSet seen; seen.reserve(kReserve);                  // hpp:113-114
    BlockReader rd(in, o.format, false, o.block_bytes);      // hpp:115,118
    Summary s;
    while (true) {
        Record r = rd.next();
        SingleKey key(r.seq(), r.seq_len - 1);               // hpp:124,131 — newline excluded
        ++s.total;
        if (seen.find(key) == seen.end()) {                  // hpp:133-138
            sink.write(r.at, r.bytes());
            seen.insert(std::move(key));
        } else {
            ++s.dups;                                        // hpp:140
        }
        if (rd.block_end()) { rd.refill(); if (rd.done()) break; }   // hpp:126-144
    }
    if (o.verbose)                                           // hpp:146-147
        std::cout << s.total << " reads processed, out of which " << s.dups << " duplicates were removed.\n";
    return s;
}

// ======================= a11: paired, ordered ===============================

// hash_dup_remover.hpp:194-255.  i-th record of file 1 pairs with the i-th of
// file 2; IDs are not compared; stops at the shorter file.
static Summary filter_paired_ordered(const std::string& in1, const std::string& in2,
                                     const std::string& out1, const std::string& out2, const Options& o)
{
    OutFile sink1(out1), sink2(out2);                        // hpp:202-203
    PairSet seen; seen.reserve(kReserve);
    BlockReader left(in1, o.format, true, o.block_bytes), right(in2, o.format, true, o.block_bytes); // hpp:208-212
    Summary s;
    while (true) {
        Record l = left.next(), r = right.next();            // hpp:232-233
        PairKey key(l.seq(), l.seq_len - 1, r.seq(), r.seq_len - 1);
        ++s.total;
        if (seen.find(key) == seen.end()) {
            sink1.write(l.at, l.bytes());
            sink2.write(r.at, r.bytes());
            seen.insert(std::move(key));
        } else {
            ++s.dups;
        }
        // hpp:228-230,249-250: the reference refreshes both sides when either
        // hits its block end.  Restated for the envelope where that is sound
        // (both sides cross block ends at the same record).
        if (left.block_end() || right.block_end()) {
            if (left.block_end())  left.refill();
            if (right.block_end()) right.refill();
            if (left.done() || right.done()) break;
        }
    }
    if (o.verbose)                                           // hpp:253-254
        std::cout << s.total << " read pairs processed, out of which " << s.dups << " duplicates were removed.\n";
    return s;
}

// ======================= a10/a12: paired, unordered =========================

struct LoadedFile {
    std::string bytes;
    std::vector<Record> recs;
};

// external_sort.hpp:88-117 reads every record and orders views by operator<
// (= tag compare).  Restated in memory: the on-disk chunk/merge machinery
// (external_sort.hpp:119-215) changes nothing for unique tags.
static void load_sorted_by_tag(const std::string& name, Format f, LoadedFile& lf)
{
    InFile file(name);
    std::vector<char> chunk(1 << 22);
    while (!file.eof()) {
        ssize_t k = file.read(chunk.data(), static_cast<ssize_t>(chunk.size()));
        lf.bytes.append(chunk.data(), static_cast<size_t>(k));
    }
    const char* b = lf.bytes.data();
    const char* e = b + lf.bytes.size();
    Record r;
    ssize_t n = parse_record(f, true, b, e, r);
    if (n < 0) throw std::runtime_error("Not enough memory to read a single object!");
    while (n >= 0) { lf.recs.push_back(r); b += n; n = parse_record(f, true, b, e, r); }
    std::stable_sort(lf.recs.begin(), lf.recs.end(),
                     [](const Record& a, const Record& c) { return compare_tags(a, c) < 0; });
}

// Merge-join positions over two tag-sorted record lists.
// tail_rule (hash_dup_remover.hpp:279-340): cursors advance only while neither
// is on its file's LAST record; then exactly one more comparison is made.
static void join_sorted(const std::vector<Record>& L, const std::vector<Record>& R, bool tail_rule,
                        std::vector<std::pair<size_t, size_t>>& hits, size_t& unmatched)
{
    size_t i = 0, j = 0;
    const size_t n = L.size(), m = R.size();
    unmatched = 0;
    if (n == 0 || m == 0) return;
    if (tail_rule) {
        while (i + 1 < n && j + 1 < m) {                     // hpp:281
            int c = compare_tags(L[i], R[j]);
            if (c < 0)      { ++i; ++unmatched; }            // hpp:284-287
            else if (c > 0) { ++j; ++unmatched; }            // hpp:288-290
            else            { hits.emplace_back(i, j); ++i; ++j; }
        }
        int c = compare_tags(L[i], R[j]);                    // hpp:317-340 "check 2 last records"
        if (c == 0) hits.emplace_back(i, j); else ++unmatched;
    } else {
        while (i < n && j < m) {
            int c = compare_tags(L[i], R[j]);
            if (c < 0)      { ++i; ++unmatched; }
            else if (c > 0) { ++j; ++unmatched; }
            else            { hits.emplace_back(i, j); ++i; ++j; }
        }
        unmatched += (n - i) + (m - j);
    }
}

// hash_dup_remover.hpp:150-192 + 257-347.
static Summary filter_paired_unordered(const std::string& in1, const std::string& in2,
                                       const std::string& out1, const std::string& out2, const Options& o)
{
    LoadedFile A, B;
    load_sorted_by_tag(in1, o.format, A);                    // hpp:161-167
    load_sorted_by_tag(in2, o.format, B);                    // hpp:169-173
    OutFile sink1(out1), sink2(out2);                        // hpp:265-266
    std::vector<std::pair<size_t, size_t>> hits;
    Summary s;
    join_sorted(A.recs, B.recs, o.reference_tail_rule, hits, s.unmatched);
    PairSet seen; seen.reserve(kReserve);
    for (auto [i, j] : hits) {                               // hpp:291-309 (tag order)
        const Record& l = A.recs[i];
        const Record& r = B.recs[j];
        PairKey key(l.seq(), l.seq_len - 1, r.seq(), r.seq_len - 1);
        ++s.total;
        if (seen.find(key) == seen.end()) {
            sink1.write(l.at, l.bytes());
            sink2.write(r.at, r.bytes());
            seen.insert(std::move(key));
        } else {
            ++s.dups;
        }
    }
    if (o.verbose) {                                         // hpp:342-346
        std::cout << s.total << " valid read pairs processed, out of which " << s.dups << " duplicates were removed.\n";
        std::cout << s.unmatched << " Non-matching entries from both files were skipped.\n";
    }
    return s;
}

Summary filter_paired(const std::string& in1, const std::string& in2,
                      const std::string& out1, const std::string& out2,
                      bool unordered, const Options& o)
{
    return unordered ? filter_paired_unordered(in1, in2, out1, out2, o)
                     : filter_paired_ordered(in1, in2, out1, out2, o);
}

} // namespace fqo

// ======================= flat C entry points ================================

static void copy_err(char* err, int64_t cap, const char* what)
{
    if (!err || cap <= 0) return;
    std::strncpy(err, what, static_cast<size_t>(cap - 1));
    err[cap - 1] = '\0';
}

extern "C" {

int64_t fqo_pack_sequence(const char* seq, int64_t len, uint64_t* out, int64_t cap)
{
    std::vector<uint64_t> w;
    try { fqo::pack_sequence(w, seq, len); }
    catch (const fqo::UnknownBase& e) { return -(1 + static_cast<int64_t>(static_cast<unsigned char>(e.ch))); }
    for (size_t i = 0; i < w.size() && static_cast<int64_t>(i) < cap; ++i) out[i] = w[i];
    return static_cast<int64_t>(w.size());
}

int64_t fqo_parse_block(const char* buf, int64_t n, int format, int want_tag,
                        int64_t* out, int64_t cap, int64_t* consumed, char* err, int64_t errcap)
{
    int64_t count = 0;
    const char* b = buf; const char* e = buf + n;
    try {
        fqo::Record r;
        while (count < cap) {
            ssize_t k = fqo::parse_record(static_cast<fqo::Format>(format), want_tag != 0, b, e, r);
            if (k < 0) break;
            int64_t* o = out + 7 * count;
            o[0] = r.at - buf; o[1] = r.id_len; o[2] = r.seq_len; o[3] = r.plus_len; o[4] = r.qual_len;
            o[5] = want_tag ? (r.tag - buf) : -1; o[6] = want_tag ? r.tag_len : -1;
            ++count; b += k;
        }
    } catch (const std::exception& ex) { copy_err(err, errcap, ex.what()); return -1; }
    if (consumed) *consumed = b - buf;
    return count;
}

int fqo_compare_tags(const char* a, int64_t alen, const char* b, int64_t blen)
{
    return fqo::compare_tag_bytes(a, alen, b, blen);
}

int64_t fqo_dedup_single(const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                         uint64_t n, uint8_t* keep, uint64_t* bad_index)
{
    fqo::SingleSet seen; seen.reserve(fqo::kReserve);
    int64_t dups = 0;
    for (uint64_t i = 0; i < n; ++i) {
        try {
            fqo::SingleKey key(reinterpret_cast<const char*>(bytes + off[i]), static_cast<ssize_t>(len[i]));
            if (seen.find(key) == seen.end()) { keep[i] = 1; seen.insert(std::move(key)); }
            else { keep[i] = 0; ++dups; }
        } catch (const fqo::UnknownBase& e) {
            if (bad_index) *bad_index = i;
            return -(1 + static_cast<int64_t>(static_cast<unsigned char>(e.ch)));
        }
    }
    return dups;
}

int64_t fqo_dedup_paired(const uint8_t* b1, const uint64_t* off1, const uint32_t* len1,
                         const uint8_t* b2, const uint64_t* off2, const uint32_t* len2,
                         uint64_t n, uint8_t* keep, uint64_t* bad_index)
{
    fqo::PairSet seen; seen.reserve(fqo::kReserve);
    int64_t dups = 0;
    for (uint64_t i = 0; i < n; ++i) {
        try {
            fqo::PairKey key(reinterpret_cast<const char*>(b1 + off1[i]), static_cast<ssize_t>(len1[i]),
                             reinterpret_cast<const char*>(b2 + off2[i]), static_cast<ssize_t>(len2[i]));
            if (seen.find(key) == seen.end()) { keep[i] = 1; seen.insert(std::move(key)); }
            else { keep[i] = 0; ++dups; }
        } catch (const fqo::UnknownBase& e) {
            if (bad_index) *bad_index = i;
            return -(1 + static_cast<int64_t>(static_cast<unsigned char>(e.ch)));
        }
    }
    return dups;
}

int64_t fqo_join_tags(const uint8_t* t1, const uint64_t* off1, const uint32_t* len1, uint64_t n1,
                      const uint8_t* t2, const uint64_t* off2, const uint32_t* len2, uint64_t n2,
                      int tail_rule, uint64_t* out_i1, uint64_t* out_i2, uint64_t* unmatched)
{
    // Records carrying only a tag; `at` is abused to remember the input index.
    auto build = [](const uint8_t* t, const uint64_t* off, const uint32_t* len, uint64_t n,
                    std::vector<fqo::Record>& v, std::vector<uint64_t>& order) {
        v.resize(n); order.resize(n);
        for (uint64_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            return fqo::compare_tag_bytes(reinterpret_cast<const char*>(t + off[a]), len[a],
                                          reinterpret_cast<const char*>(t + off[b]), len[b]) < 0; });
        for (uint64_t k = 0; k < n; ++k) {
            v[k].tag = reinterpret_cast<const char*>(t + off[order[k]]);
            v[k].tag_len = len[order[k]];
        }
    };
    std::vector<fqo::Record> L, R; std::vector<uint64_t> ol, orr;
    build(t1, off1, len1, n1, L, ol);
    build(t2, off2, len2, n2, R, orr);
    std::vector<std::pair<size_t, size_t>> hits; size_t un = 0;
    fqo::join_sorted(L, R, tail_rule != 0, hits, un);
    for (size_t k = 0; k < hits.size(); ++k) { out_i1[k] = ol[hits[k].first]; out_i2[k] = orr[hits[k].second]; }
    if (unmatched) *unmatched = un;
    return static_cast<int64_t>(hits.size());
}

int fqo_filter_single(const char* in, const char* out, int format, int verbose,
                      int64_t block_bytes, uint64_t* total, uint64_t* dups, char* err, int64_t errcap)
{
    try {
        fqo::Options o; o.format = static_cast<fqo::Format>(format); o.verbose = verbose != 0;
        if (block_bytes > 0) o.block_bytes = block_bytes;
        fqo::Summary s = fqo::filter_single(in, out, o);
        if (total) *total = s.total;
        if (dups) *dups = s.dups;
        return 0;
    } catch (const std::exception& ex) { copy_err(err, errcap, ex.what()); return 1; }
}

int fqo_filter_paired(const char* in1, const char* in2, const char* out1, const char* out2,
                      int format, int unordered, int tail_rule, int verbose, int64_t block_bytes,
                      uint64_t* total, uint64_t* dups, uint64_t* unmatched, char* err, int64_t errcap)
{
    try {
        fqo::Options o; o.format = static_cast<fqo::Format>(format); o.verbose = verbose != 0;
        o.reference_tail_rule = tail_rule != 0;
        if (block_bytes > 0) o.block_bytes = block_bytes;
        fqo::Summary s = fqo::filter_paired(in1, in2, out1, out2, unordered != 0, o);
        if (total) *total = s.total;
        if (dups) *dups = s.dups;
        if (unmatched) *unmatched = s.unmatched;
        return 0;
    } catch (const std::exception& ex) { copy_err(err, errcap, ex.what()); return 1; }
}

} // extern "C"
