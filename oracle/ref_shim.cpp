// ref_shim.cpp — flat C entry points onto the REFERENCE'S OWN object code.
// TEST INFRASTRUCTURE ONLY (see fqd_oracle.hpp).
//
// oracle/Makefile compiles /root/reference/src/{seq_utils,fastqview,fastaview}.cpp
// where they lie (they include no Boost) and links them with this file into
// oracle/_ref/libfqd_ref.so.  Nothing of the reference is copied: this file
// only #includes its headers through -I/root/reference/src and forwards calls,
// so ctypes can reach the C++ symbols.  It exists to pin the restatement in
// fqd_oracle.cpp and to generate tests/golden/*.json (tests/golden/make_golden.py).
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "seq_utils.hpp"     // reference: SeqUtils::pattern2number, seq2hash
#include "fastqview.hpp"     // reference: FastqView, FastqViewWithId
#include "fastaview.hpp"     // reference: FastaView, FastaViewWithId

namespace {
// Walks buf with the reference's read_new (fastqview.cpp:89-119,
// fastaview.cpp:75-93).  out gets 5 int64 per record:
// {start, id_len, seq_len, size, seq_offset}.  Returns the record count, or
// -1 when the reference throws.
template <class View>
static int64_t walk(char* buf, int64_t n, int64_t* out, int64_t cap, int64_t* consumed)
{
    int64_t count = 0;
    char* b = buf; char* e = buf + n;
    try {
        while (count < cap) {
            View v;
            std::streamsize k = v.read_new(b, e);
            if (k < 0) break;
            int64_t* o = out + 5 * count;
            o[0] = v.start() - buf; o[1] = v.id_len(); o[2] = v.seq_len(); o[3] = v.size(); o[4] = v.seq() - buf;
            ++count; b += k;
        }
    } catch (const std::exception&) { return -1; }
    if (consumed) *consumed = b - buf;
    return count;
}

// FastqViewWithId::cmp / FastaViewWithId::cmp (fastqview.cpp:168-204,
// fastaview.cpp:131-167) on two single records; returns sign(cmp), or 99 when
// either record does not parse.
template <class View>
static int cmp_records(char* a, int64_t an, char* b, int64_t bn)
{
    View va, vb;
    try {
        if (va.read_new(a, a + an) < 0 || vb.read_new(b, b + bn) < 0) return 99;
    } catch (const std::exception&) { return 99; }
    int c = va.cmp(vb);
    return (c > 0) - (c < 0);
}
} // namespace

extern "C" {

// SeqUtils::seq2hash (seq_utils.cpp:35-49).  Returns the chunk count, or -1
// when the reference throws (unknown base).
int64_t ref_seq2hash(const char* seq, int64_t len, uint64_t* out, int64_t cap)
{
    std::vector<uint64_t> h;
    try { SeqUtils::seq2hash(h, seq, static_cast<ssize_t>(len)); }
    catch (const std::exception&) { return -1; }
    for (size_t i = 0; i < h.size() && static_cast<int64_t>(i) < cap; ++i) out[i] = h[i];
    return static_cast<int64_t>(h.size());
}

// SeqUtils::pattern2number (seq_utils.cpp:23-33); *ok = 0 when it throws.
uint64_t ref_pattern2number(const char* seq, int64_t len, int* ok)
{
    try { *ok = 1; return SeqUtils::pattern2number(seq, static_cast<size_t>(len)); }
    catch (const std::exception&) { *ok = 0; return 0; }
}

int64_t ref_walk_fastq(char* buf, int64_t n, int64_t* out, int64_t cap, int64_t* consumed)
{ return walk<FastqView>(buf, n, out, cap, consumed); }
int64_t ref_walk_fasta(char* buf, int64_t n, int64_t* out, int64_t cap, int64_t* consumed)
{ return walk<FastaView>(buf, n, out, cap, consumed); }

int ref_cmp_fastq_ids(char* a, int64_t an, char* b, int64_t bn) { return cmp_records<FastqViewWithId>(a, an, b, bn); }
int ref_cmp_fasta_ids(char* a, int64_t an, char* b, int64_t bn) { return cmp_records<FastaViewWithId>(a, an, b, bn); }

} // extern "C"
