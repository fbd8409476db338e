// scan_check — reads a text from stdin and prints what the host record scanner makes of it,
// once scanning from the front and once per requested thread count with the parallel scanner:
//   "<threads> <consumed> <n records> <fail set> <fail what>|<fnv of all record fields>"
// tests/test_host_scan.py asserts that every line after the first says the same as the first.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>
#include "../../fastq-dupaway_amd/host/records.hpp"

using namespace fqdhost;

static void report(unsigned threads, size_t consumed, const std::vector<RecordRef>& recs, const ParseFailure& fail)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { for (int k = 0; k < 8; ++k) { h ^= (v >> (8 * k)) & 0xFF; h *= 1099511628211ull; } };
    for (const RecordRef& r : recs) { mix(r.start); mix(r.size); mix(r.id_len); mix(r.seq_len); mix(r.tag_off); mix(r.tag_len); }
    std::printf("%u %zu %zu %d %s|%s|%llu\n", threads, consumed, recs.size(), int(fail.set), fail.what.c_str(),
                fail.diag.substr(0, fail.diag.find('\n')).c_str(), static_cast<unsigned long long>(h));
}

int main(int argc, char** argv)
{
    const Format f = (argc > 1 && std::string(argv[1]) == "fasta") ? Format::Fasta : Format::Fastq;
    const bool want_tag = argc > 2 && std::atoi(argv[2]) != 0;
    std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    {
        std::vector<RecordRef> recs; ParseFailure fail;
        const size_t c = scan_records(f, want_tag, text.data(), text.size(), recs, fail);
        report(1, c, recs, fail);
    }
    for (unsigned t = 2; t <= 7; ++t) {
        std::vector<RecordRef> recs; ParseFailure fail;
        const size_t c = scan_records_parallel(f, want_tag, text.data(), text.size(), recs, fail, t);
        report(1, c, recs, fail);
    }
    return 0;
}
