// The device BGZF coder's logic (fastq-dupaway_amd/csrc/fqd_bgzf_core.hpp) run thread by thread on the
// CPU, phase by phase as the kernels of fqd_bgzf.hip run it between barriers: input file -> BGZF file.
// Test infrastructure only (tests/test_bgzf_core.py inflates the result with Python's gzip).
//   bgzf_core_check <in> <out.gz> <lines_per_record>     prints: members stored_members bytes_out
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../fastq-dupaway_amd/csrc/fqd_bgzf_core.hpp"

using namespace fqd::bgzf;

struct HostOr { void operator()(uint32_t* p, uint32_t v) const { *p |= v; } };

struct Lines {
    std::vector<uint16_t> ls = std::vector<uint16_t>(kMaxLines + 2);
    uint32_t line_at[kThreads];
    uint32_t n_lines = 0;
    bool on;
};

static void index_lines(const uint8_t* data, uint32_t L, Lines& x)
{
    uint32_t total = 0;
    for (uint32_t t = 0; t < kThreads; ++t) {
        uint32_t lo, hi; chunk_of(t, L, lo, hi);
        x.line_at[t] = total;
        const Scan sc = scan_chunk(Linear{data}, lo, hi);
        total += uint32_t(__builtin_popcountll(sc.nl.lo) + __builtin_popcountll(sc.nl.hi));
    }
    x.n_lines = total;
    x.on = total <= kMaxLines;
    x.ls[0] = 0;
    if (!x.on) return;
    for (uint32_t t = 0; t < kThreads; ++t) {
        uint32_t lo, hi; chunk_of(t, L, lo, hi);
        const Scan sc = scan_chunk(Linear{data}, lo, hi);
        uint32_t k = x.line_at[t] + 1;
        for (uint64_t m = sc.nl.lo; m; m &= m - 1) x.ls[k++] = uint16_t(lo + uint32_t(__builtin_ctzll(m)) + 1);
        for (uint64_t m = sc.nl.hi; m; m &= m - 1) x.ls[k++] = uint16_t(lo + 64 + uint32_t(__builtin_ctzll(m)) + 1);
    }
}

struct Counter {
    uint64_t* hist;
    void literal(uint32_t b) { ++hist[b]; }
    void match(uint32_t len, uint32_t dist) { ++hist[length_symbol(len).sym]; ++hist[kLitLen + dist_symbol(dist).sym]; }
};

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint32_t K = uint32_t(std::atoi(argv[3]));
    const uint64_t n = in.size(), members = (n + kMember - 1) / kMember;
    in.resize(n + 64);
    std::vector<uint64_t> hist(kLitLen + kDist, 0);
    Lines x;
    for (uint64_t m = 0; m < members; m += sample_every(members)) {
        const uint8_t* data = in.data() + m * kMember;
        const uint32_t L = uint32_t(std::min<uint64_t>(kMember, n - m * kMember));
        index_lines(data, L, x);
        for (uint32_t t = 0; t < kThreads; ++t) {
            uint32_t lo, hi; chunk_of(t, L, lo, hi);
            Counter c{hist.data()};
            const Scan sc = scan_chunk(Linear{data}, lo, hi);
            const Columns col = column_masks(Linear{data}, lo, hi, x.ls.data(), x.line_at[t], x.n_lines, L, x.on, K);
            parse_chunk(Linear{data}, lo, hi, sc, col, c);
        }
    }
    static Codes codes;
    build_codes(hist.data(), members, codes);
    std::FILE* out = std::fopen(argv[2], "wb");
    uint64_t bytes_out = 0, stored_members = 0;
    std::vector<uint32_t> slot(kSlot / 4);
    for (uint64_t m = 0; m < members; ++m) {
        const uint8_t* data = in.data() + m * kMember;
        const uint32_t L = uint32_t(std::min<uint64_t>(kMember, n - m * kMember));
        std::fill(slot.begin(), slot.end(), 0u);
        index_lines(data, L, x);
        uint32_t bits[kThreads], before[kThreads], body = 0;
        for (uint32_t t = 0; t < kThreads; ++t) {
            uint32_t lo, hi; chunk_of(t, L, lo, hi);
            BitCounter price{codes.lit, codes.dist};
            const Scan sc = scan_chunk(Linear{data}, lo, hi);
            const Columns col = column_masks(Linear{data}, lo, hi, x.ls.data(), x.line_at[t], x.n_lines, L, x.on, K);
            parse_chunk(Linear{data}, lo, hi, sc, col, price);
            bits[t] = price.bits; before[t] = body; body += bits[t];
        }
        const uint32_t total_bits = codes.header_bits + body + (codes.lit[256] >> 16);
        uint32_t clen = (total_bits + 7) / 8;
        const bool stored = clen >= L + 5;
        if (stored) { clen = L + 5; ++stored_members; }
        HostOr orw;
        uint32_t crc[kThreads];
        for (uint32_t t = 0; t < kThreads; ++t) {
            uint32_t lo, hi; chunk_of(t, L, lo, hi);
            if (!stored) {
                BitWriter<HostOr> w(slot.data(), kHeadBytes * 8 + (t == 0 ? 0 : codes.header_bits + before[t]), orw);
                if (t == 0)
                    for (uint32_t at = 0; at < codes.header_bits; at += 32)
                        w.put(codes.header_bits - at >= 32 ? codes.header[at >> 5] : codes.header[at >> 5] & ((1u << (codes.header_bits - at)) - 1u),
                              codes.header_bits - at >= 32 ? 32 : codes.header_bits - at);
                Emitter<HostOr> emit{codes.lit, codes.dist, w};
                const Scan sc = scan_chunk(Linear{data}, lo, hi);
                const Columns col = column_masks(Linear{data}, lo, hi, x.ls.data(), x.line_at[t], x.n_lines, L, x.on, K);
                parse_chunk(Linear{data}, lo, hi, sc, col, emit);
                if (t == kThreads - 1) w.put(codes.lit[256] & 0xFFFFu, codes.lit[256] >> 16);
                w.finish();
            } else {
                if (t == 0) { BitWriter<HostOr> w(slot.data(), kHeadBytes * 8, orw); w.put(1, 8); w.put(L, 16); w.put(~L & 0xFFFFu, 16); w.finish(); }
                BitWriter<HostOr> w(slot.data(), (kHeadBytes + 5 + lo) * 8, orw);
                for (uint32_t p = lo; p < hi; ++p) w.put(data[p], 8);
                w.finish();
            }
            crc[t] = crc_chunk(codes.crc_table, Linear{data}, lo, hi);
        }
        for (uint32_t k = 0; k < kLevels; ++k)
            for (uint32_t t = 0; t < kThreads; t += 2u << k) crc[t] = crc_advance(codes.crc_shift[k], crc[t]) ^ crc[t + (1u << k)];
        const uint32_t total = kHeadBytes + clen + kTailBytes;
        BitWriter<HostOr> h(slot.data(), 0, orw);
        h.put(31u | (139u << 8) | (8u << 16) | (4u << 24), 32); h.put(0, 32); h.put(0u | (255u << 8) | (6u << 16), 32);
        h.put(uint32_t('B') | (uint32_t('C') << 8) | (2u << 16), 32); h.put(total - 1, 16); h.finish();
        BitWriter<HostOr> tl(slot.data(), (kHeadBytes + clen) * 8, orw);
        tl.put(crc[0] ^ 0xFFFFFFFFu, 32); tl.put(L, 32); tl.finish();
        std::fwrite(slot.data(), 1, total, out);
        bytes_out += total;
    }
    static const unsigned char eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::fwrite(eof, 1, sizeof eof, out);
    std::fclose(out);
    std::printf("%llu %llu %llu\n", (unsigned long long)members, (unsigned long long)stored_members, (unsigned long long)(bytes_out + sizeof eof));
    return 0;
}
