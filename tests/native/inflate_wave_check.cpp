// The GPU BGZF reader's per-wave decoder (fastq-dupaway_amd/csrc/fqd_inflate_wave.hpp) run on the CPU: the lanes
// of the wave are a loop, a phase ends when the loop does.  Every member of a BGZF file is inflated with it and the
// result written out, for tests/test_inflate_core.py to compare with what zlib makes of the same file.  Test
// infrastructure only.
//   inflate_wave_check <in.gz> <out> [lanes: 64 | 8]      prints: members bad_members bytes_out
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <vector>

#include "../../fastq-dupaway_amd/csrc/fqd_inflate_wave.hpp"

using namespace fqd::winf;

template <uint32_t L>
struct LoopCtx {
    static constexpr uint32_t kLanes = L;
    template <class F> void lanes(F f) { for (uint32_t l = 0; l < L; ++l) f(l); }
    template <class F> void lanes_open(F f) { for (uint32_t l = 0; l < L; ++l) f(l); }
    void sync() {}
    template <class F> uint64_t ballot(F f) { uint64_t m = 0; for (uint32_t l = 0; l < L; ++l) m |= uint64_t(f(l) ? 1u : 0u) << l; return m; }
    uint32_t same(uint32_t v) const { return v; }
    void add(uint32_t* p, uint32_t v) { *p += v; }
    void mark(int) {}
};

template <uint32_t L>
static int run(const std::vector<uint8_t>& in, size_t n, const char* out_name)
{
    std::FILE* out = std::fopen(out_name, "wb");
    unsigned long long members = 0, bad = 0, bytes = 0;
    std::vector<uint8_t> buf((1 << 16) + 16, 0xEE);
    auto sh = std::make_unique<Shared<L>>();
    std::vector<Token> tok(kTokenRoom);
    LoopCtx<L> ctx;
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
        // a private copy of the member's bytes, rounded out to whole words: what lies beyond must not be read as data
        std::vector<uint32_t> words((total - 26 + 3) / 4 + 2, 0xFFFFFFFFu);
        uint8_t* cp = reinterpret_cast<uint8_t*>(words.data()) + (members % 4);
        std::memcpy(cp, p + 18, total - 26);
        const uint32_t st = inflate_member(ctx, *sh, cp, uint32_t(total - 26), dst, isize, tok.data());
        if (st == kOk) for (size_t k = 0; k < buf.size(); ++k)
            if ((buf.data() + k < dst || buf.data() + k >= dst + isize) && buf[k] != 0xEE) { std::fprintf(stderr, "member at %zu wrote outside its bytes\n", at); ++bad; break; }
        if (st != kOk) { ++bad; std::fprintf(stderr, "member at %zu: status %u\n", at, st); }
        else { std::fwrite(dst, 1, isize, out); bytes += isize; }
        at += total;
    }
    std::fclose(out);
    std::printf("%llu %llu %llu\n", members, bad, bytes);
#if defined(FQD_WINF_STATS)
    const Stats& s = stats();
    std::fprintf(stderr, "blocks %llu windows %llu rounds %llu lane_decodes %llu | tokens %llu groups %llu group_rounds %llu cut %llu | waits (per turn): one periodic %llu, one partly %llu, several %llu\n",
                 s.blocks, s.windows, s.rounds, s.lane_decodes, s.tokens, s.groups, s.group_rounds, s.cut, s.dep_one_periodic, s.dep_one_partial, s.dep_many);
#endif
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t n = in.size();
    in.resize(n + 16);
    const int lanes = argc > 3 ? std::atoi(argv[3]) : 64;
    return lanes == 8 ? run<8>(in, n, argv[2]) : run<64>(in, n, argv[2]);
}
