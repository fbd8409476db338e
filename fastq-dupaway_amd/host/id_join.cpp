#include "id_join.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace fqdhost {

void load_whole_file(const std::string& name, Format f, size_t block_bytes, LoadedFile& out)
{
    InputFile file(name);
    std::vector<char> carry;
    std::vector<RecordRef> refs;
    bool first = true;
    while (true) {
        const size_t cap = std::max(block_bytes, carry.size() + block_bytes / 2);
        std::unique_ptr<PinnedBuffer> buf(new PinnedBuffer());
        buf->reserve(cap + 16);
        size_t have = carry.size();
        if (have) std::memcpy(buf->p, carry.data(), have);
        carry.clear();
        have += file.read(buf->p + have, cap - have, host_threads());
        refs.clear();
        const size_t consumed = scan_records_parallel(f, true, buf->p, have, refs, out.failure, host_threads());
        if (first && refs.empty() && !out.failure.set)
            throw std::runtime_error("Not enough memory to read a single object!");
        first = false;
        const uint32_t chunk = static_cast<uint32_t>(out.chunks.size());
        for (const RecordRef& r : refs)
            out.recs.push_back(FileRecord{buf->p + r.start, r.size, r.id_len, r.seq_len, r.tag_off, r.tag_len, chunk});
        const bool at_end = file.eof();
        if (!out.failure.set && !at_end) {
            if (refs.empty()) throw std::runtime_error("Not enough memory to read a single object!");
            carry.assign(buf->p + consumed, buf->p + have);
        }
        out.chunk_used.push_back(consumed);
        out.chunks.push_back(std::move(buf));
        if (out.failure.set || at_end) break;
    }
}

TailOutcome full_join_outcome(const JoinLookup& j)
{
    if (j.n == 0 || j.m == 0) return TailOutcome{0, false, 0};     // nothing is compared at all
    return TailOutcome{j.n_pairs, false, j.n + j.m - 2 * j.n_pairs};
}

TailOutcome reference_tail_rule(const JoinLookup& j)
{
    const uint64_t n = j.n, m = j.m;
    if (n == 0 || m == 0) return TailOutcome{0, false, 0};          // the loop and the last check never run
    if (n == 1 || m == 1) {
        // the loop body never runs: one comparison of the two smallest tags (hpp:317-340)
        const bool equal = j.match_a(0) == 0;
        return TailOutcome{equal ? 1u : 0u, !equal && j.n_pairs > 0, equal ? 0u : 1u};
    }
    // The full merge visits its states (i, j) in order; the reference's loop is that merge cut at
    // the first state with i == n-1 or j == m-1.  i becomes n-1 right after L[n-2] is consumed, in
    // the state (n-2, j1): j1 = its partner's position when it has one, else everything of file 2
    // with a tag <= its own has been consumed by then.  If file 2's cursor is still below m-1
    // there (j1 <= m-2) that is where the loop stops; otherwise file 2 reached its last record
    // first and the same holds with the files swapped.
    uint64_t i_exit, j_exit;
    const uint32_t mx = j.match_a(n - 2);
    const uint64_t j1 = mx != kNoPartner ? mx : j.count_b_le_a(n - 2);
    if (j1 <= m - 2) { i_exit = n - 1; j_exit = j1 + (mx != kNoPartner ? 1 : 0); }
    else {
        const uint32_t my = j.match_b(m - 2);
        const uint64_t i2 = my != kNoPartner ? my : j.count_a_le_b(m - 2);
        j_exit = m - 1; i_exit = i2 + (my != kNoPartner ? 1 : 0);
    }
    // pairs below both last records are all found before the stop; a pair that involves a last
    // record is the last pair of the full join, and found only if the stop state is that pair
    const bool special = j.match_a(n - 1) != kNoPartner || j.match_b(m - 1) != kNoPartner;
    const uint64_t before = j.n_pairs - (special ? 1 : 0);
    const bool last_equal = j.match_a(i_exit) == j_exit;            // equal tags in a visited state = a pair of the full join
    TailOutcome out;
    out.pairs = before + (last_equal ? 1 : 0);
    out.drop_last = special && !last_equal;
    out.unmatched = i_exit + j_exit - 2 * before + (last_equal ? 0 : 1);
    return out;
}

} // namespace fqdhost
