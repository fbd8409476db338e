// id_join.hpp — the `--unordered` read-ID join (hash_dup_remover.hpp:150-192,257-347).
// The reference sorts both files by ID tag on disk (ExternalSorter, external_sort.hpp:88-215)
// and merge-joins them; here both files are indexed in memory and joined on the same key
// with the same ordering (FastqViewWithId::cmp, fastqview.cpp:168-204).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "records.hpp"

namespace fqdhost {

struct FileRecord {
    const char* text;        // first byte of the record (stable for the life of the LoadedFile)
    uint32_t size, id_len, seq_len, tag_off, tag_len;
    uint32_t chunk;          // which chunk of the LoadedFile holds it
    const char* tag() const { return text + tag_off; }
};

struct LoadedFile {
    std::vector<std::unique_ptr<PinnedBuffer>> chunks;   // page-locked: uploaded to HBM as they are
    std::vector<size_t> chunk_used;                      // bytes of each chunk that hold records
    std::vector<FileRecord> recs;        // in file order
    ParseFailure failure;                // a malformed record ended the load
};

// Reads and indexes a whole file.  An empty file (or one whose first record is incomplete)
// throws "Not enough memory to read a single object!" like the reference's sorter does.
void load_whole_file(const std::string& name, Format f, size_t block_bytes, LoadedFile& out);

// The two device primitives the join is built on (include/fqdupaway.h: fqd_sort_tags,
// fqd_match_sorted_tags), bound to an engine by the caller.
struct fqd_engine_fwd;
struct TagJoinDevice {
    // perm[k] = record index of the k-th smallest tag
    virtual void sort(const LoadedFile& f, std::vector<uint32_t>& perm) = 0;
    // match[k] = position in perm_b of the record whose tag equals a's perm_a[k], or 0xFFFFFFFF
    virtual void match(const LoadedFile& a, const std::vector<uint32_t>& perm_a,
                       const LoadedFile& b, const std::vector<uint32_t>& perm_b, std::vector<uint32_t>& match) = 0;
    virtual ~TagJoinDevice() = default;
};

// Joins on the ID tag.  `pairs` receives (index in a, index in b) in tag order; unmatched
// counts skipped records the way the reference does.  tail_rule: stop as the reference's
// merge loop does, as soon as either side is on its LAST record, then compare once more
// (hash_dup_remover.hpp:279-340).
void join_by_tag(const LoadedFile& a, const LoadedFile& b, bool tail_rule, TagJoinDevice& dev,
                 std::vector<std::pair<uint64_t, uint64_t>>& pairs, uint64_t& unmatched);

} // namespace fqdhost
