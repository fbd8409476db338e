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

} // namespace fqd_plan
