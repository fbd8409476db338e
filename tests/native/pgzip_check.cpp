// The parallel reader of ordinary gzip files (fastq-dupaway_amd/host/pgzip.hpp) on its own: inflates a file with it and
// writes the text out, for tests/test_pgzip.py to compare with what zlib makes of the same file.  Test infrastructure.
//   pgzip_check <in.gz> <out> <threads> [read size]      prints: bytes_out, or "corrupt" and exits 3
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../fastq-dupaway_amd/host/pgzip.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const int fd = ::open(argv[1], O_RDONLY);
    if (fd < 0) return 2;
    struct stat st;
    ::fstat(fd, &st);
    std::FILE* out = std::fopen(argv[2], "wb");
    const size_t want = argc > 4 ? size_t(std::atoll(argv[4])) : size_t(1) << 20;
    unsigned long long total = 0;
    try {
        fqdhost::pgz::Reader r(fd, uint64_t(st.st_size), unsigned(std::atoi(argv[3])));
        std::vector<char> buf(want);
        for (;;) {
            const size_t k = r.read(buf.data(), buf.size());
            if (k == 0) break;
            std::fwrite(buf.data(), 1, k, out);
            total += k;
        }
    } catch (const std::invalid_argument& e) { std::printf("unsupported %s\n", e.what()); return 4; }
    catch (const std::exception& e) { std::fclose(out); std::printf("corrupt\n"); return 3; }
    std::fclose(out);
    std::printf("%llu\n", total);
    return 0;
}
