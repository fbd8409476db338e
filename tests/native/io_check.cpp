// io_check — exercises the host driver's file layer (fastq-dupaway_amd/host/file_io.cpp) without a GPU:
//   io_check w <out> <n bytes> <seed>       writes n pseudo-random bytes in uneven pieces; prints "<n> <fnv>"
//   io_check r <in> <chunk> <threads>       reads the file back in chunks; prints "<n> <fnv>"
//   io_check m <members.gz> <out.gz> <threads>   copies the finished members of a BGZF file (minus its end marker) into a
//                                           new .gz file through write_members, between two ordinary write() calls
// tests/test_host_io.py compares the two, also against Python's gzip module.
#include "../../fastq-dupaway_amd/host/file_io.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
using namespace fqdhost;
int main(int argc, char** argv) {
    std::string mode = argv[1];
    try {
    if (mode == "w") {
        size_t n = std::strtoull(argv[3], nullptr, 10); unsigned seed = std::atoi(argv[4]);
        std::string data(n, 'A'); for (size_t i = 0; i < n; ++i) { seed = seed * 1664525u + 1013904223u; data[i] = "ACGTN\n@+I"[(seed >> 24) % 9]; }
        OutputFile o(argv[2]); size_t at = 0; while (at < n) { size_t k = std::min<size_t>(n - at, 1 + (seed = seed * 1664525u + 1013904223u) % 200000); o.write(data.data() + at, k); at += k; } o.close();
        unsigned long long h = 1469598103934665603ull; for (unsigned char c : data) { h ^= c; h *= 1099511628211ull; }
        std::printf("%zu %llu\n", n, h);
    } else if (mode == "m") {
        InputFile raw(argv[2], true);
        std::vector<char> all(size_t(1) << 30); size_t n = 0;
        while (!raw.eof()) { const size_t k = raw.read(all.data() + n, all.size() - n, 1); if (!k) break; n += k; }
        n -= 28;                                                   // the end-of-file member is the writer's own to add
        OutputFile o(argv[3]);
        o.write("@before\n", 8);
        o.write_members(all.data(), n / 2 / 4 * 4, unsigned(std::atoi(argv[4])));     // (any split works as long as whole members go in order:
        o.write_members(all.data() + n / 2 / 4 * 4, 0, 1);                                //  here the bytes are simply passed on in two calls)
        o.write_members(all.data() + n / 2 / 4 * 4, n - n / 2 / 4 * 4, unsigned(std::atoi(argv[4])));
        o.write("@after\n", 7);
        o.close();
        std::printf("%zu 0\n", n);
    } else {
        InputFile in(argv[2]); size_t chunk = std::strtoull(argv[3], nullptr, 10); unsigned th = std::atoi(argv[4]);
        std::vector<char> buf(chunk); unsigned long long h = 1469598103934665603ull; size_t tot = 0;
        while (!in.eof()) { size_t k = in.read(buf.data(), chunk, th); for (size_t i = 0; i < k; ++i) { h ^= (unsigned char)buf[i]; h *= 1099511628211ull; } tot += k; if (k == 0 && in.eof()) break; }
        std::printf("%zu %llu\n", tot, h);
    }
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
