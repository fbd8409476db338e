// The GPU reader of ORDINARY gzip files (fastq-dupaway_amd/csrc/fqd_gunzip_core.hpp) run on the CPU, unit by unit as the
// kernels of fqd_gunzip.hip run it: block starts guessed per unit, every unit decoded into 16-bit symbols from its guess,
// the chain of unit ends and starts checked, windows made unit after unit, symbols turned into bytes.  What comes out is
// written for tests/test_gunzip_core.py to compare with what zlib makes of the same file.  Test infrastructure only.
//   gunzip_core_check <in.gz> <out> <unit_bytes> [max_ratio] [serial|planes]     prints: status units bytes_out deflate_bytes
//   serial: fqd_gunzip_core.hpp's one decoder per unit writing symbols; planes (what the kernels do): the wave decoder of
//   fqd_inflate_wave.hpp (its lanes a loop here) run per unit over TWO made-up windows at once, whose two texts together ARE the symbols
//   status: ok | chain (a unit does not start where the one before it ended) | bad (damaged data) | full (a unit's room) | header
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <vector>

#include "../../fastq-dupaway_amd/csrc/fqd_gunzip_core.hpp"
#include "../../fastq-dupaway_amd/csrc/fqd_inflate_wave.hpp"

using namespace fqd::gunz;

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

struct ArraySink {
    uint16_t* p; uint64_t cap, n = 0;
    bool room(uint32_t need) const { return n + need <= cap; }
    void put(uint16_t s) { p[n++] = s; }
    void copy(uint32_t d, uint32_t len)
    {
        const int64_t from = int64_t(n) - int64_t(d);
        for (uint32_t k = 0; k < len; ++k) { const int64_t i = from + int64_t(k % d); p[n + k] = i < 0 ? uint16_t(256 + int64_t(kWindow) + i) : p[i]; }
        n += len;
    }
    uint64_t count() const { return n; }
};

// The member's header (RFC 1952): where the deflate stream starts; 0: not a header this reader takes.
static size_t deflate_start(const std::vector<uint8_t>& in, size_t n)
{
    if (n < 18 || in[0] != 31 || in[1] != 139 || in[2] != 8) return 0;
    const uint8_t flg = in[3];
    size_t at = 10;
    if (flg & 4) { if (at + 2 > n) return 0; at += 2 + (in[at] | (size_t(in[at + 1]) << 8)); }
    if (flg & 8) { while (at < n && in[at]) ++at; ++at; }
    if (flg & 16) { while (at < n && in[at]) ++at; ++at; }
    if (flg & 2) at += 2;
    return at < n ? at : 0;
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t n = file.size();
    const uint64_t unit_bytes = std::strtoull(argv[3], nullptr, 10);
    const uint64_t ratio = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 16;
    std::FILE* out = std::fopen(argv[2], "wb");
    const size_t ds = deflate_start(file, n);
    if (!ds || unit_bytes < 64) { std::printf("header 0 0 0\n"); std::fclose(out); return 0; }
    // the deflate stream and everything behind it (the trailer, further members) in a buffer of its own: aligned words, 16
    // readable bytes behind — and at an odd byte offset within the words, as it lies in a file
    const size_t len = n - ds;
    std::vector<uint64_t> words((len + 3) / 8 + 4, 0);
    uint8_t* base = reinterpret_cast<uint8_t*>(words.data()) + 3;
    std::memcpy(base, file.data() + ds, len);
    BitIn in; in.words = words.data(); in.lead = 24; in.nbits = uint64_t(len) * 8;

    const uint64_t n_units = (len + unit_bytes - 1) / unit_bytes;
    auto tables = std::make_unique<Tables>();
    uint8_t lens[320];
    // 1. guessed starts
    std::vector<uint64_t> start(n_units, UINT64_MAX);
    start[0] = 0;
    for (uint64_t u = 1; u < n_units; ++u) {
        const uint64_t lo = u * unit_bytes * 8, hi = std::min<uint64_t>((u + 1) * unit_bytes * 8, in.nbits);
        for (uint64_t p = lo; p < hi; ++p) if (plausible_block_start(in, p, lens)) { start[u] = p; break; }
    }
    // 2. every unit with a start, decoded on its own
    struct Unit { uint64_t start, end, stop_bit, span; uint32_t status; std::vector<uint16_t> sym; };
    std::vector<Unit> units;
    const bool planes = argc > 5 && !std::strcmp(argv[5], "planes");
    auto wsh = std::make_unique<fqd::winf::Shared<64>>();
    std::vector<fqd::winf::Token> wtok(fqd::winf::kTokenRoom);
    auto decode = [&](Unit& x) {
        if (planes) {
            // the unit through the byte decoder into two texts, the 32 KiB before each made up: plane_P[w] = w & 255, plane_Q[w] = (w & 255) ^ (1 + (w >> 8)).
            // A byte that comes out the same in both is a literal of the stream; one that differs was copied from place w of the window.
            const uint64_t cap = x.span * ratio + 1024;
            std::vector<uint8_t> pl[2];
            uint32_t info[2][3] = {{0, 0, 0}, {0, 0, 0}}, st[2] = {0, 0};
            const uint64_t byte0 = x.start >> 3;
            const uint32_t first_bit = uint32_t(x.start & 7u);
            const uint64_t rel_stop = x.stop_bit == UINT64_MAX ? 0xFFFFFFFFull : x.stop_bit - byte0 * 8;
            const uint32_t comp_len = uint32_t(std::min<uint64_t>(len - byte0, 1u << 28));
            std::vector<uint32_t> cw(comp_len / 4 + 4, 0);                   // the unit's bytes on, at the alignment they have in the file
            uint8_t* cp = reinterpret_cast<uint8_t*>(cw.data()) + (byte0 & 3u);
            std::memcpy(cp, base + byte0, comp_len);
            LoopCtx<64> ctx;
            for (int p = 0; p < 2; ++p) {
                pl[p].assign(kWindow + cap + 16, 0xEE);
                for (uint32_t w = 0; w < kWindow; ++w) pl[p][w] = p ? uint8_t((w & 255u) ^ (1u + (w >> 8))) : uint8_t(w & 255u);
            }
            // ONE decode writes both planes (as gz_decode_planes_kernel does): literals stored twice, matches copied in each
            st[0] = st[1] = fqd::winf::inflate_stretch(ctx, *wsh, cp, comp_len, first_bit, uint32_t(std::min<uint64_t>(rel_stop, 0xFFFFFFFFull)),
                                                       pl[0].data(), kWindow, uint32_t(kWindow + cap), wtok.data(), info[0], pl[1].data());
            info[1][0] = info[0][0]; info[1][1] = info[0][1]; info[1][2] = info[0][2];
            if (st[0] != fqd::winf::kOk || st[1] != fqd::winf::kOk || info[0][0] != info[1][0] || info[0][1] != info[1][1]) {
                x.status = st[0] == fqd::winf::kOutputOverrun || st[1] == fqd::winf::kOutputOverrun ? uint32_t(kOutputFull) : uint32_t(kBadData);
                x.end = 0; x.sym.clear();
                return;
            }
            x.end = byte0 * 8 + info[0][0];
            x.status = info[0][2] == 2u ? uint32_t(kFinal) : uint32_t(kBoundary);
            const uint64_t n = info[0][1] - kWindow;
            x.sym.resize(n);
            for (uint64_t i = 0; i < n; ++i) {
                const uint32_t a = pl[0][kWindow + i], b = pl[1][kWindow + i];
                x.sym[i] = a == b ? uint16_t(a) : uint16_t(256u + (a | (((a ^ b) - 1u) << 8)));
            }
            return;
        }
        x.sym.assign(x.span * ratio + 1024, 0);
        ArraySink sink{x.sym.data(), x.sym.size()};
        State st; st.pos = st.start_bit = x.start;
        while (st.status == kOk) decode_some(in, *tables, lens, st, x.stop_bit, sink, 1000);      // in stretches, as the GPU does
        x.end = st.pos; x.status = st.status;
        x.sym.resize(sink.n);
    };
    for (uint64_t u = 0; u < n_units; ++u) {
        if (start[u] == UINT64_MAX) continue;
        Unit x; x.start = start[u];
        uint64_t next = u + 1;
        while (next < n_units && start[next] == UINT64_MAX) ++next;          // a unit without a start belongs to the one before it
        x.stop_bit = next < n_units ? next * unit_bytes * 8 : UINT64_MAX;
        x.span = (next - u) * unit_bytes;
        decode(x);
        units.push_back(std::move(x));
    }
    // a guess that did not hold: the unit is decoded again from where the one before it really ended (fqd_gunzip.hip does the same)
    unsigned repairs = 0;
    for (size_t k = 1; k < units.size(); ++k) {
        if (units[k - 1].status == kFinal) { units.resize(k); break; }       // what follows is the trailer
        if (units[k - 1].status == kBoundary && units[k - 1].end != units[k].start && repairs < 64) { units[k].start = units[k - 1].end; decode(units[k]); ++repairs; }
    }
    if (repairs) std::fprintf(stderr, "%u units decoded again\n", repairs);
    // 3. the chain
    const char* verdict = "ok";
    for (size_t k = 0; k < units.size(); ++k) {
        const Unit& x = units[k];
        if (x.status == kOutputFull) { verdict = "full"; break; }
        if (x.status == kBadData || x.status == kInputEnd) { verdict = "bad"; break; }
        if (k + 1 < units.size() ? (x.status != kBoundary || x.end != units[k + 1].start) : x.status != kFinal) { verdict = "chain"; break; }
    }
    unsigned long long bytes = 0;
    if (!std::strcmp(verdict, "ok")) {
        // 4 + 5. windows unit after unit, then the bytes
        std::vector<uint8_t> win(kWindow, 0), next(kWindow, 0), text;
        for (const Unit& x : units) {
            text.resize(x.sym.size());
            for (size_t i = 0; i < x.sym.size(); ++i) text[i] = x.sym[i] < 256 ? uint8_t(x.sym[i]) : win[x.sym[i] - 256];
            if (!text.empty()) std::fwrite(text.data(), 1, text.size(), out);
            bytes += text.size();
            for (uint32_t k = 0; k < kWindow; ++k) next[k] = window_byte(win.data(), x.sym.data(), x.sym.size(), k);
            win.swap(next);
        }
    }
    std::fclose(out);
    const unsigned long long deflate_bytes = units.empty() ? 0 : (units.back().end + 7) / 8;
    std::printf("%s %zu %llu %llu\n", verdict, units.size(), bytes, deflate_bytes);
    return 0;
}
