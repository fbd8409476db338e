#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../fastq-dupaway_amd/host/pgzip.hpp"
using namespace fqdhost::pgz;
int main(int argc, char** argv) {
    int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st);
    void* m = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    BitIn in; in.base = (const uint8_t*)m; in.nbytes = st.st_size;
    for (int rep = 0; rep < 3; ++rep) {
        Piece p; p.sym.reserve(300u << 20);
        auto t0 = std::chrono::steady_clock::now();
        decode_piece(in, 10 * 8, UINT64_MAX, p);     // (gzip.compress writes a 10-byte header)
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("decode: %zu symbols in %.3f s = %.0f MB/s (stop %d)\n", p.sym.size(), dt, p.sym.size() / dt / 1e6, int(p.stop));
        std::vector<uint8_t> w, out(p.sym.size());
        t0 = std::chrono::steady_clock::now();
        resolve(p.sym, 0, p.sym.size(), w, out.data());
        double dr = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        unsigned long c = crc32_z(0, out.data(), out.size());
        double dc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("resolve %.3f s = %.0f MB/s; crc %.3f s = %.0f MB/s (%lx)\n", dr, out.size() / dr / 1e6, dc, out.size() / dc / 1e6, c);
    }
}
