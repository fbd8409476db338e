// The GPU BGZF reader's per-thread decoder (fastq-dupaway_amd/csrc/fqd_inflate_core.hpp) run on the
// CPU: every member of a BGZF file is inflated with it and the result written out, for
// tests/test_inflate_core.py to compare with what zlib makes of the same file.  Test infrastructure only.
//   inflate_core_check <in.gz> <out>      prints: members bad_members bytes_out
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../fastq-dupaway_amd/csrc/fqd_inflate_core.hpp"

using namespace fqd::inflate;

struct Lens {
    uint8_t v[kLitSymbols + kDistSymbols + 2];
    uint32_t get(uint32_t i) const { return v[i]; }
    void set(uint32_t i, uint8_t x) { v[i] = x; }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t n = in.size();
    in.resize(n + 16);
    std::FILE* out = std::fopen(argv[2], "wb");
    unsigned long long members = 0, bad = 0, bytes = 0;
    std::vector<uint8_t> buf((1 << 16) + 16, 0xEE);
    alignas(4) static uint8_t table_bytes[kPackedBytes];
    PackedTables<1> t(table_bytes, 0); Lens l;
    for (size_t at = 0; at + 18 <= n;) {
        const uint8_t* p = in.data() + at;
        if (!(p[0] == 31 && p[1] == 139 && p[2] == 8 && p[3] == 4 && p[12] == 'B' && p[13] == 'C')) { ++bad; break; }
        const size_t total = (p[16] | (size_t(p[17]) << 8)) + 1;
        const uint8_t* tail = p + total - 8;
        const uint32_t isize = tail[4] | (uint32_t(tail[5]) << 8) | (uint32_t(tail[6]) << 16) | (uint32_t(tail[7]) << 24);
        ++members;
        if (isize > (1u << 16)) { ++bad; at += total; continue; }
        // members land at every alignment, as they do in the text of a file; the bytes around them are not theirs
        uint8_t* dst = buf.data() + 4 + (members & 3);
        std::fill(buf.begin(), buf.end(), uint8_t(0xEE));
        const uint32_t st = inflate_member(p + 18, uint32_t(total - 26), dst, isize, t, l);
        if (st == kOk) for (size_t k = 0; k < buf.size(); ++k)
            if ((buf.data() + k < dst || buf.data() + k >= dst + isize) && buf[k] != 0xEE) { std::fprintf(stderr, "member at %zu wrote outside its bytes\n", at); ++bad; break; }
        if (st != kOk) { ++bad; std::fprintf(stderr, "member at %zu: status %u\n", at, st); }
        else { std::fwrite(dst, 1, isize, out); bytes += isize; }
        at += total;
    }
    std::fclose(out);
    std::printf("%llu %llu %llu\n", members, bad, bytes);
    return 0;
}
