// id_join.hpp — the `--unordered` read-ID join (hash_dup_remover.hpp:150-192,257-347).
// The reference sorts both files by ID tag on disk (ExternalSorter, external_sort.hpp:88-215)
// and merge-joins them; here both files are indexed in memory and joined on the same key
// with the same ordering (FastqViewWithId::cmp, fastqview.cpp:168-204).
#pragma once
#include <cstdint>
#include <functional>
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

// What the device join (include/fqdupaway.h: fqd_join_tags) found: the FULL inner join of the two
// files on the ID tag, k-th record with a tag in file 1 paired with the k-th in file 2.  Positions
// are positions in each file's tag order.  The arrays stay on the device; these accessors fetch
// what the end-of-file rule needs (a handful of entries, and a tag search only when the
// second-to-last record of a file has no partner).
constexpr uint32_t kNoPartner = 0xFFFFFFFFu;
struct JoinLookup {
    uint64_t n = 0, m = 0;               // records in file 1 / file 2
    uint64_t n_pairs = 0;                // pairs of the full join
    std::function<uint32_t(uint64_t)> match_a;        // sorted position in file 1 -> partner's sorted position in file 2
    std::function<uint32_t(uint64_t)> match_b;        // the reverse
    std::function<uint64_t(uint64_t)> count_b_le_a;   // records of file 2 whose tag is <= the tag at sorted position i of file 1
    std::function<uint64_t(uint64_t)> count_a_le_b;   // records of file 1 whose tag is <= the tag at sorted position j of file 2
};

// The reference's merge loop (hash_dup_remover.hpp:279-340) advances only while NEITHER cursor is
// on its file's last record, then compares once more and stops.  Against the full
// join that can only lose the LAST pair in tag order — the one that involves a file's last
// record — which survives iff the loop happens to stop exactly on it.
struct TailOutcome {
    uint64_t pairs;        // pairs the reference processes: n_pairs, or n_pairs - 1 when the last one is lost
    bool     drop_last;    // the last pair of the full join is not processed
    uint64_t unmatched;    // what the reference counts as "Non-matching entries"
};
TailOutcome reference_tail_rule(const JoinLookup& j);
TailOutcome full_join_outcome(const JoinLookup& j);

} // namespace fqdhost
