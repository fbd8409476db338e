// fqd_oracle — CPU restatement of fastq-dupaway's hash-based `--fast` path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
// `cpu_baseline` leg and __graft_entry__.smoke() may build, load or run
// anything in oracle/.  The shipped engine (fastq-dupaway_amd/) never links,
// loads or falls back to it.
//
// Every function cites the reference file:line it restates (paths are under
// /root/reference/src/).  It is a restatement in our own structure (one
// runtime-dispatched driver instead of eight template instantiations, an
// in-memory ID sort instead of the on-disk chunk sorter), not a copy.
//
// Pinning status (see tests/test_oracle_*.py):
//   * packing (a1-a3), record parsing (a14) and ID-tag ordering (a13) are
//     checked against the reference's OWN seq_utils.cpp / fastqview.cpp /
//     fastaview.cpp compiled where they lie into oracle/_ref/ (those three
//     files need no Boost);
//   * the drivers (a9-a12) are checked byte-for-byte against every fixture the
//     reference's tests hold for this path (test/inputs -> test/expected,
//     committed as data under tests/golden/reference_fixtures/).
//   The full reference binary needs Boost 1.81, which this image lacks: it is
//   unbuildable here and no stand-in for Boost is written.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <sys/types.h>
#include <vector>

namespace fqo {

// ---- a1-a3: base-5 / 17-mer packing -------------------------------------
constexpr long kChunkBases = 17;                 // seq_utils.hpp:9 CHUNKSIZE

struct UnknownBase : std::runtime_error {        // seq_utils.cpp:17-19
    char ch;
    explicit UnknownBase(char c)
        : std::runtime_error("Supported sequence character set: {A, N, C, G, T}!"), ch(c) {}
};

int      base_digit(char c);                                        // seq_utils.cpp:3-21
uint64_t pack_chunk(const char* s, size_t n);                       // seq_utils.cpp:23-33
void     pack_sequence(std::vector<uint64_t>& out, const char* s, ssize_t len); // seq_utils.cpp:35-49

// ---- a4-a7: keys, equality, bucket hash ---------------------------------
struct SingleKey {                                                  // hash_dup_remover.hpp:19-29
    ssize_t len = 0;
    std::vector<uint64_t> words;
    SingleKey() = default;
    SingleKey(const char* s, ssize_t n);                            // hash_dup_remover.cpp:4-8
    bool operator==(const SingleKey& o) const;                      // hash_dup_remover.cpp:10-14
};
struct PairKey {                                                    // hash_dup_remover.hpp:31-41
    ssize_t llen = 0, rlen = 0;
    std::vector<uint64_t> lwords, rwords;
    PairKey() = default;
    PairKey(const char* l, ssize_t ln, const char* r, ssize_t rn);  // hash_dup_remover.cpp:16-24
    bool operator==(const PairKey& o) const;                        // hash_dup_remover.cpp:26-33
};
struct SingleKeyHash { size_t operator()(const SingleKey&) const; };// hash_dup_remover.hpp:43-52
struct PairKeyHash   { size_t operator()(const PairKey&)   const; };// hash_dup_remover.hpp:54-68

// ---- a13/a14: record views ----------------------------------------------
enum Format { FASTQ = 0, FASTA = 1 };

struct Record {                      // fastqview.hpp:25-27,40-42 ; fastaview.hpp:27-29,44-45
    const char* at = nullptr;        // first byte of the ID line
    ssize_t id_len = 0, seq_len = 0, plus_len = 0, qual_len = 0;   // each includes its '\n'
    const char* tag = nullptr;       // join key (only when parsed with want_tag)
    ssize_t tag_len = 0;
    ssize_t bytes() const { return id_len + seq_len + plus_len + qual_len; }
    const char* seq() const { return at + id_len; }
    bool empty() const { return at == nullptr; }
};

// Parses one record at [b, e).  Returns its byte size, or -1 when the block
// ends before the record does.  Throws on a bad first byte or a seq/qual
// length mismatch, after printing the reference's diagnostic to stderr.
// fastqview.cpp:89-138,190-204 ; fastaview.cpp:75-100,153-167
ssize_t parse_record(Format f, bool want_tag, const char* b, const char* e, Record& r);
int     compare_tags(const Record& a, const Record& b);            // fastqview.cpp:168-178

// ---- a15/a16: block reader, plain/gz files --------------------------------
class InFile;    // file_utils.hpp:25-57 ; file_utils.cpp:53-79
class OutFile;   // file_utils.hpp:71-79 ; file_utils.cpp:83-92

constexpr ssize_t kDefaultBlock = 5L * 100L * 1024L * 1024L;  // hash_dup_remover.hpp:115 (5*HUNDRED_MB)

// ---- a9-a12: drivers -------------------------------------------------------
struct Summary { size_t total = 0, dups = 0, unmatched = 0; };

struct Options {
    Format  format = FASTQ;
    bool    verbose = false;
    ssize_t block_bytes = kDefaultBlock;
    // true  = the reference's merge-join exactly, including its end-of-file
    //         rule (hash_dup_remover.hpp:281,317-340);
    // false = the intended full inner join.
    bool    reference_tail_rule = true;
};

Summary filter_single(const std::string& in, const std::string& out, const Options&);   // hpp:105-148
Summary filter_paired(const std::string& in1, const std::string& in2,
                      const std::string& out1, const std::string& out2,
                      bool unordered, const Options&);                                   // hpp:150-347

} // namespace fqo

// ---- flat C entry points for ctypes (tests / bench cpu_baseline) ----------
extern "C" {
// Packs one sequence; returns chunk count, or -(1+bad_byte) on an unknown base.
int64_t fqo_pack_sequence(const char* seq, int64_t len, uint64_t* out, int64_t cap);
// Parses consecutive records in buf[0,n); fills 7 int64 per record
// {start,id_len,seq_len,plus_len,qual_len,tag_off,tag_len}; returns count,
// -1 on a format error (message copied to err).
int64_t fqo_parse_block(const char* buf, int64_t n, int format, int want_tag,
                        int64_t* out, int64_t cap, int64_t* consumed, char* err, int64_t errcap);
int     fqo_compare_tags(const char* a, int64_t alen, const char* b, int64_t blen);
// First-occurrence exact dedup over packed ASCII reads.  keep[i]=1 iff read i
// is the first with its sequence.  Returns #duplicates, or -(1+bad_byte) with
// *bad_index set on an unknown base.
int64_t fqo_dedup_single(const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                         uint64_t n, uint8_t* keep, uint64_t* bad_index);
int64_t fqo_dedup_paired(const uint8_t* b1, const uint64_t* off1, const uint32_t* len1,
                         const uint8_t* b2, const uint64_t* off2, const uint32_t* len2,
                         uint64_t n, uint8_t* keep, uint64_t* bad_index);
// ID join of two tag lists (bytes+off+len each).  Writes matched (i1,i2) pairs
// in tag order; returns the pair count.  tail_rule as in Options.
int64_t fqo_join_tags(const uint8_t* t1, const uint64_t* off1, const uint32_t* len1, uint64_t n1,
                      const uint8_t* t2, const uint64_t* off2, const uint32_t* len2, uint64_t n2,
                      int tail_rule, uint64_t* out_i1, uint64_t* out_i2, uint64_t* unmatched);
// File drivers; return 0, or 1 with the what() text in err.
int fqo_filter_single(const char* in, const char* out, int format, int verbose,
                      int64_t block_bytes, uint64_t* total, uint64_t* dups, char* err, int64_t errcap);
int fqo_filter_paired(const char* in1, const char* in2, const char* out1, const char* out2,
                      int format, int unordered, int tail_rule, int verbose, int64_t block_bytes,
                      uint64_t* total, uint64_t* dups, uint64_t* unmatched, char* err, int64_t errcap);
}
