// fqd_shard_plan.hpp — where things lie in the fixed-size-slab exchange of csrc/fqd_shard.hip.  Host-only, no HIP:
// tests/native/shard_plan_check.cpp plays whole rounds (overflows included) with these functions on the CPU, and
// the gloo test of tests/test_shard_plan.py moves real bytes between processes by them.
//
// A rank's SEND buffer (fqd_partition_slabs): `world` slabs of `cap` slots — slab d holds the first cap keys bound
// for owner d — then the spill region: the keys the slabs had no room for, owner after owner.
// An owner RECEIVES slab s of every source at slot s*cap of the room at its key store's tail; a spill from source s
// (exactly sized: both ends of a pair know its true count) arrives in a buffer of the owner's own, behind a copy of
// the slabs, source after source; the owner then lays the round out compactly, source after source, each source's
// slab part followed by its spill.
// FLAGS travel back in the same shape: `cap` bytes per pair (a short slab's tail means nothing), plus the spill's.
#pragma once
#include <cstdint>

namespace fqd_plan {

inline uint64_t over(uint64_t count, uint64_t cap) { return count > cap ? count - cap : 0; }

// First slot of part `part`'s slab in a send buffer / of source `part`'s slab in the owner's receive room.
inline uint64_t slab_slot(uint32_t part, uint64_t cap) { return uint64_t(part) * cap; }

// First slot of part `part`'s spill: in a send buffer with counts = what this rank sends to every owner; in an
// owner's spill buffer with counts = what it receives from every source.
inline uint64_t spill_slot(const uint64_t* counts, uint32_t world, uint32_t part, uint64_t cap)
{
    uint64_t at = uint64_t(world) * cap;
    for (uint32_t p = 0; p < part; ++p) at += over(counts[p], cap);
    return at;
}

// Did this owner receive a spill (so that it lays the round out again, compactly)?
inline bool owner_is_compact(const uint64_t* in_counts, uint32_t world, uint64_t cap)
{
    for (uint32_t p = 0; p < world; ++p) if (in_counts[p] > cap) return true;
    return false;
}

// Where source `src`'s records (and so its flags) start at the owner once the round is inserted.
inline uint64_t owner_offset(const uint64_t* in_counts, uint32_t world, uint32_t src, uint64_t cap)
{
    if (!owner_is_compact(in_counts, world, cap)) return slab_slot(src, cap);
    uint64_t at = 0;
    for (uint32_t p = 0; p < src; ++p) at += in_counts[p];
    return at;
}

// Records the owner inserts for the round (unused slab slots included when it keeps the slab layout).
inline uint64_t owner_records(const uint64_t* in_counts, uint32_t world, uint64_t cap)
{
    if (!owner_is_compact(in_counts, world, cap)) return uint64_t(world) * cap;
    uint64_t n = 0;
    for (uint32_t p = 0; p < world; ++p) n += in_counts[p];
    return n;
}

// ---- slabs cut into sub-slabs ---------------------------------------------------------------------------------
// A rank's batch of a round is cut into `chunks` chunks of chunk_reads consecutive reads and every slab into one
// sub-slab of sub_cap slots per chunk (slab capacity cap = chunks * sub_cap): chunk c's keys for owner d lie, in input
// order, in sub-slab (d, c).  That is what lets the encoder write every key once, straight to its place, with no
// workgroup waiting for another (csrc/fqd_kernels.hpp, encode_chunks).  A slab filled from its first slot on (the
// three-step grouping) is the same thing seen as full, partial and empty sub-slabs: classic_count below.
struct Geometry {
    uint32_t world = 1, chunks = 1;
    uint64_t chunk_reads = 0, sub_cap = 0;
    uint64_t cap() const { return uint64_t(chunks) * sub_cap; }
};

// round_reads: most reads a rank brings to a round.  About a thousand chunks (whole 256-read tiles, at least 4096
// reads) so that every workgroup of the chip finds one; a sub-slab holds a fair share of a chunk plus 5.5 standard
// deviations of a binomial share (about one sub-slab in 10^7 overflows; the round is then grouped again, exactly).
// forced_cap (tests): one chunk, one sub-slab of that many slots.
inline Geometry geometry(uint64_t round_reads, uint32_t world, uint64_t forced_cap = 0)
{
    Geometry g;
    g.world = world ? world : 1;
    if (round_reads == 0) round_reads = 1;
    if (forced_cap) { g.chunks = 1; g.chunk_reads = (round_reads + 255) / 256 * 256; g.sub_cap = forced_cap; return g; }
    uint64_t chunk = (round_reads + 1023) / 1024;
    chunk = (chunk + 255) / 256 * 256;
    if (chunk < 4096) chunk = 4096;
    g.chunk_reads = chunk;
    g.chunks = uint32_t((round_reads + chunk - 1) / chunk);
    if (g.world == 1) { g.sub_cap = chunk; return g; }
    const uint64_t fair = (chunk + g.world - 1) / g.world;
    uint64_t sd = 0; while ((sd + 1) * (sd + 1) <= fair) ++sd;
    g.sub_cap = (fair + (11 * sd + 1) / 2 + 16 + 7) & ~7ull;
    return g;
}

// What sub-slab c of a slab holds when `total` keys were written to the slab from its first slot on; the last
// sub-slab's count carries whatever the slab had no room for (total > cap: that much spilled).
inline uint64_t classic_count(uint64_t total, uint32_t c, uint32_t chunks, uint64_t sub_cap)
{
    const uint64_t before = uint64_t(c) * sub_cap;
    uint64_t v = total > before ? total - before : 0;
    if (c + 1 < chunks && v > sub_cap) v = sub_cap;
    return v;
}

// Sub-slabs a source takes up at an owner that lays the round out again because somebody spilled: its slab, and its
// spill as further sub-slabs behind it.
inline uint64_t owner_sub_slabs(uint64_t total, const Geometry& g)
{
    return g.chunks + (total > g.cap() ? (total - g.cap() + g.sub_cap - 1) / g.sub_cap : 0);
}

} // namespace fqd_plan
