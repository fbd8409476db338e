// bgzf_pack <in> <out.gz> — copies a file through the host driver's file layer, i.e. writes it as
// BGZF on worker threads (tools/e2e_bench.py uses it to make large compressed inputs quickly).
#include <cstdio>
#include <vector>
#include "../fastq-dupaway_amd/host/file_io.hpp"

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: bgzf_pack <in> <out.gz>\n"); return 2; }
    try {
        fqdhost::InputFile in(argv[1]);
        fqdhost::OutputFile out(argv[2]);
        std::vector<char> buf(64u << 20);
        while (!in.eof()) {
            const size_t k = in.read(buf.data(), buf.size(), 8);
            if (k) out.write(buf.data(), k);
        }
        out.close();
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
