// shard_plan_check — the fixed-size-slab exchange of csrc/fqd_shard.hip played on the CPU with the very functions
// (csrc/fqd_shard_plan.hpp) that place its slabs, spills and flags: W ranks, random owners, slabs small enough to
// overflow now and then.  Checks that every owner sees its records in (source rank, position) order and that every
// record's flag finds its way back to the position the record came from.
//   usage: shard_plan_check <ranks> <seed> [cap]        -> "ok <records> <spilled>" or a message and exit 1
// Built as a shared object too (tests/test_shard_plan.py): the extern "C" functions at the end hand the same
// geometry to the gloo test, which moves real bytes between processes by it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../fastq-dupaway_amd/csrc/fqd_shard_plan.hpp"

using namespace fqd_plan;

namespace {

struct Rank {
    std::vector<uint64_t> key;          // input order: (rank << 32) | position
    std::vector<uint32_t> owner;
    std::vector<uint64_t> grouped;      // send buffer: slabs, then the spill region
    std::vector<uint32_t> origin;
    std::vector<uint64_t> out_counts, in_counts;
    std::vector<uint64_t> slot;         // owner side: the slabs as received
    std::vector<uint64_t> spill;        // owner side: copy of the slabs + the spills
    std::vector<uint64_t> inserted;     // what the owner inserts, in order
    std::vector<uint8_t> keep_recv, keep_back, keep;
};

uint8_t flag_of(uint64_t key) { return uint8_t(((key * 0x9E3779B97F4A7C15ull) >> 40) & 1u); }

} // namespace

#ifndef SHARD_PLAN_NO_MAIN
int main(int argc, char** argv)
{
    const uint32_t W = argc > 1 ? uint32_t(std::atoi(argv[1])) : 4;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    const uint64_t n_max = 200 + rng() % 400;
    const uint64_t cap = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 16 + rng() % (n_max / W + 40);
    std::vector<Rank> r(W);
    uint64_t total = 0, spilled = 0;
    // ---- every rank groups its records by owner: fqd_partition_slabs ------------------------------------------
    for (uint32_t s = 0; s < W; ++s) {
        const uint64_t n = rng() % 7 == 0 ? 0 : 1 + rng() % n_max;
        const bool skew = rng() % 3 == 0;
        r[s].key.resize(n); r[s].owner.resize(n); r[s].keep.assign(n, 9);
        for (uint64_t i = 0; i < n; ++i) { r[s].key[i] = (uint64_t(s) << 32) | i; r[s].owner[i] = (skew && rng() % 2) ? 0u : uint32_t(rng() % W); }
        r[s].out_counts.assign(W, 0);
        for (uint64_t i = 0; i < n; ++i) r[s].out_counts[r[s].owner[i]]++;
        const uint64_t slots = spill_slot(r[s].out_counts.data(), W, W, cap);
        r[s].grouped.assign(slots, ~0ull); r[s].origin.assign(slots, 0xFFFFFFFFu);
        std::vector<uint64_t> seen(W, 0);
        for (uint64_t i = 0; i < n; ++i) {                      // stable: input order within a part
            const uint32_t d = r[s].owner[i];
            const uint64_t local = seen[d]++;
            const uint64_t at = local < cap ? slab_slot(d, cap) + local : spill_slot(r[s].out_counts.data(), W, d, cap) + (local - cap);
            r[s].grouped[at] = r[s].key[i]; r[s].origin[at] = uint32_t(i);
        }
        total += n;
        for (uint32_t d = 0; d < W; ++d) spilled += over(r[s].out_counts[d], cap);
    }
    // ---- forward all-to-all: fixed-size slabs, counts beside them ----------------------------------------------
    for (uint32_t d = 0; d < W; ++d) { r[d].slot.assign(uint64_t(W) * cap, ~0ull); r[d].in_counts.assign(W, 0); }
    for (uint32_t s = 0; s < W; ++s)
        for (uint32_t d = 0; d < W; ++d) {
            for (uint64_t i = 0; i < cap; ++i) r[d].slot[slab_slot(s, cap) + i] = r[s].grouped[slab_slot(d, cap) + i];
            r[d].in_counts[s] = r[s].out_counts[d];
        }
    // ---- spills: exactly sized, both ends know the count -------------------------------------------------------
    for (uint32_t d = 0; d < W; ++d) {
        if (!owner_is_compact(r[d].in_counts.data(), W, cap)) continue;
        r[d].spill.assign(spill_slot(r[d].in_counts.data(), W, W, cap), ~0ull);
        for (uint64_t i = 0; i < uint64_t(W) * cap; ++i) r[d].spill[i] = r[d].slot[i];
    }
    for (uint32_t s = 0; s < W; ++s)
        for (uint32_t d = 0; d < W; ++d) {
            const uint64_t c = r[s].out_counts[d];
            if (c <= cap) continue;
            const uint64_t from = spill_slot(r[s].out_counts.data(), W, d, cap), to = spill_slot(r[d].in_counts.data(), W, s, cap);
            for (uint64_t i = 0; i < c - cap; ++i) r[d].spill[to + i] = r[s].grouped[from + i];
        }
    // ---- owners insert: slab layout, or the round laid out again ------------------------------------------------
    for (uint32_t d = 0; d < W; ++d) {
        const bool compact = owner_is_compact(r[d].in_counts.data(), W, cap);
        const uint64_t n_ins = owner_records(r[d].in_counts.data(), W, cap);
        r[d].inserted.assign(n_ins, ~0ull);
        for (uint32_t s = 0; s < W; ++s) {
            const uint64_t c = r[d].in_counts[s], head = c < cap ? c : cap, at = owner_offset(r[d].in_counts.data(), W, s, cap);
            for (uint64_t i = 0; i < head; ++i) r[d].inserted[at + i] = (compact ? r[d].spill : r[d].slot)[slab_slot(s, cap) + i];
            for (uint64_t i = 0; i + cap < c; ++i) r[d].inserted[at + cap + i] = r[d].spill[spill_slot(r[d].in_counts.data(), W, s, cap) + i];
        }
        // (source rank, position) order, no record lost or doubled
        uint64_t last = 0; bool first = true; uint64_t real = 0;
        for (uint64_t k : r[d].inserted) {
            if (k == ~0ull) continue;
            if (!first && k <= last) { std::printf("owner %u: records out of order\n", d); return 1; }
            last = k; first = false; ++real;
        }
        uint64_t expect = 0; for (uint32_t s = 0; s < W; ++s) expect += r[d].in_counts[s];
        if (real != expect) { std::printf("owner %u: %llu records, expected %llu\n", d, (unsigned long long)real, (unsigned long long)expect); return 1; }
        r[d].keep_recv.assign(n_ins + cap, 7);
        for (uint64_t i = 0; i < n_ins; ++i) r[d].keep_recv[i] = r[d].inserted[i] == ~0ull ? 5 : flag_of(r[d].inserted[i]);
    }
    // ---- flags back: cap bytes a pair, plus the spill's ---------------------------------------------------------
    for (uint32_t s = 0; s < W; ++s) r[s].keep_back.assign(r[s].grouped.size() + cap, 3);
    for (uint32_t d = 0; d < W; ++d)
        for (uint32_t s = 0; s < W; ++s) {
            const uint64_t c = r[d].in_counts[s], from = owner_offset(r[d].in_counts.data(), W, s, cap);
            for (uint64_t i = 0; i < cap; ++i) r[s].keep_back[slab_slot(d, cap) + i] = r[d].keep_recv[from + i];
            for (uint64_t i = 0; i + cap < c; ++i) r[s].keep_back[spill_slot(r[s].out_counts.data(), W, d, cap) + i] = r[d].keep_recv[from + cap + i];
        }
    for (uint32_t s = 0; s < W; ++s) {
        for (uint64_t at = 0; at < r[s].grouped.size(); ++at) if (r[s].origin[at] != 0xFFFFFFFFu) r[s].keep[r[s].origin[at]] = r[s].keep_back[at];
        for (uint64_t i = 0; i < r[s].key.size(); ++i)
            if (r[s].keep[i] != flag_of(r[s].key[i])) { std::printf("rank %u record %llu: wrong flag %u\n", s, (unsigned long long)i, r[s].keep[i]); return 1; }
    }
    std::printf("ok %llu %llu\n", (unsigned long long)total, (unsigned long long)spilled);
    return 0;
}
#endif

extern "C" {
uint64_t plan_over(uint64_t count, uint64_t cap) { return over(count, cap); }
uint64_t plan_slab_slot(uint32_t part, uint64_t cap) { return slab_slot(part, cap); }
uint64_t plan_spill_slot(const uint64_t* counts, uint32_t world, uint32_t part, uint64_t cap) { return spill_slot(counts, world, part, cap); }
int      plan_owner_is_compact(const uint64_t* in_counts, uint32_t world, uint64_t cap) { return owner_is_compact(in_counts, world, cap) ? 1 : 0; }
uint64_t plan_owner_offset(const uint64_t* in_counts, uint32_t world, uint32_t src, uint64_t cap) { return owner_offset(in_counts, world, src, cap); }
uint64_t plan_owner_records(const uint64_t* in_counts, uint32_t world, uint64_t cap) { return owner_records(in_counts, world, cap); }
void     plan_geometry(uint64_t round_reads, uint32_t world, uint64_t forced_cap, uint64_t* chunk_reads, uint32_t* chunks, uint64_t* sub_cap)
{ const Geometry g = geometry(round_reads, world, forced_cap); *chunk_reads = g.chunk_reads; *chunks = g.chunks; *sub_cap = g.sub_cap; }
uint64_t plan_classic_count(uint64_t total, uint32_t c, uint32_t chunks, uint64_t sub_cap) { return classic_count(total, c, chunks, sub_cap); }
uint64_t plan_owner_sub_slabs(uint64_t total, uint64_t chunk_reads, uint32_t chunks, uint64_t sub_cap)
{ Geometry g; g.chunk_reads = chunk_reads; g.chunks = chunks; g.sub_cap = sub_cap; return owner_sub_slabs(total, g); }
}
