// exchange_check — the offset arithmetic of the C++ multi-GPU exchange (fastq-dupaway_amd/host/
// multi_gpu.cpp: ExchangePlan, forward_transfers, backward_transfers) without a GPU: the transfers are
// executed with memcpy on host buffers for random per-pair counts and checked against the definition:
// owner d receives, in source-rank order, exactly the records each source grouped for d, and the
// per-record results travel back to the positions the records were sent from.
//   usage: exchange_check <ranks> <seed>   -> prints "ok <total records>" or a mismatch and exits 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../fastq-dupaway_amd/host/multi_gpu.hpp"

using namespace fqdhost;

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 2;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 1);
    ExchangePlan plan(n);
    for (uint64_t& c : plan.send) c = rng() % 5 == 0 ? 0 : rng() % 300;        // some empty messages
    plan.finish();
    const size_t item = 24;                                                     // bytes per record on the wire
    // source s groups its records by destination; record payload = (s, d, k)
    std::vector<std::vector<uint64_t>> grouped(n), received(n);
    std::vector<std::vector<uint8_t>> result_at_owner(n), result_at_source(n);
    for (int s = 0; s < n; ++s) {
        grouped[s].resize(plan.n_send[s] * 3);
        for (int d = 0; d < n; ++d)
            for (uint64_t k = 0; k < plan.send[size_t(s) * n + d]; ++k) {
                uint64_t* rec = &grouped[s][(plan.send_off[size_t(s) * n + d] + k) * 3];
                rec[0] = uint64_t(s); rec[1] = uint64_t(d); rec[2] = k;
            }
        result_at_source[s].assign(plan.n_send[s], 0xEE);
    }
    for (int d = 0; d < n; ++d) { received[d].assign(plan.n_recv[d] * 3, ~0ull); result_at_owner[d].resize(plan.n_recv[d]); }
    std::vector<const void*> src(n); std::vector<void*> dst(n);
    for (int r = 0; r < n; ++r) { src[r] = grouped[r].data(); dst[r] = received[r].data(); }
    uint64_t moved = 0;
    for (const Transfer& t : forward_transfers(plan, src, dst, item)) { std::memcpy(t.dst, t.src, t.bytes); moved += t.bytes / item; }
    uint64_t total = 0;
    for (int d = 0; d < n; ++d) {
        uint64_t at = 0;
        for (int s = 0; s < n; ++s)
            for (uint64_t k = 0; k < plan.send[size_t(s) * n + d]; ++k, ++at) {
                const uint64_t* rec = &received[d][at * 3];
                if (rec[0] != uint64_t(s) || rec[1] != uint64_t(d) || rec[2] != k) { std::printf("forward mismatch at owner %d record %llu\n", d, (unsigned long long)at); return 1; }
                result_at_owner[d][at] = uint8_t((s * 31 + d * 7 + k) & 0xFF);     // the owner's verdict for this record
            }
        if (at != plan.n_recv[d]) { std::printf("owner %d: %llu records, plan says %llu\n", d, (unsigned long long)at, (unsigned long long)plan.n_recv[d]); return 1; }
        total += at;
    }
    if (moved != total) { std::printf("moved %llu of %llu\n", (unsigned long long)moved, (unsigned long long)total); return 1; }
    for (int r = 0; r < n; ++r) { src[r] = result_at_owner[r].data(); dst[r] = result_at_source[r].data(); }
    for (const Transfer& t : backward_transfers(plan, src, dst, 1)) std::memcpy(t.dst, t.src, t.bytes);
    for (int s = 0; s < n; ++s)
        for (int d = 0; d < n; ++d)
            for (uint64_t k = 0; k < plan.send[size_t(s) * n + d]; ++k)
                if (result_at_source[s][plan.send_off[size_t(s) * n + d] + k] != uint8_t((s * 31 + d * 7 + k) & 0xFF)) {
                    std::printf("backward mismatch at source %d -> %d record %llu\n", s, d, (unsigned long long)k); return 1; }
    std::printf("ok %llu\n", (unsigned long long)total);
    return 0;
}
