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

void join_by_tag(const LoadedFile& a, const LoadedFile& b, bool tail_rule, TagJoinDevice& dev,
                 std::vector<std::pair<uint64_t, uint64_t>>& pairs, uint64_t& unmatched)
{
    pairs.clear(); unmatched = 0;
    const size_t n = a.recs.size(), m = b.recs.size();
    if (n == 0 || m == 0) return;
    // the sort phase (hpp:161-173) and the equality tests of the merge (hpp:283-309) run on the GPU
    std::vector<uint32_t> oa, ob, match;
    dev.sort(a, oa);
    dev.sort(b, ob);
    dev.match(a, oa, b, ob, match);
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    auto tag_a = [&](size_t k) -> const FileRecord& { return a.recs[oa[k]]; };
    auto tag_b = [&](size_t k) -> const FileRecord& { return b.recs[ob[k]]; };
    auto cmp_ab = [&](size_t i, size_t j) { return compare_tags(tag_a(i).tag(), tag_a(i).tag_len, tag_b(j).tag(), tag_b(j).tag_len); };

    if (!tail_rule) {                                        // intended semantics: full inner join
        for (size_t k = 0; k < n; ++k)
            if (match[k] != kNone) pairs.emplace_back(oa[k], ob[match[k]]);
        unmatched = n + m - 2 * pairs.size();
        return;
    }
    // The reference's loop (hpp:279-340) advances only while NEITHER cursor is on its file's last
    // record, then compares once more.  Everything it finds before that point is exactly the set
    // of matches below both second-to-last tags; what remains is one comparison at the exit state.
    if (n == 1 || m == 1) {
        if (cmp_ab(0, 0) == 0) pairs.emplace_back(oa[0], ob[0]); else unmatched = 1;
        return;
    }
    size_t i_exit, j_exit;
    const int c = cmp_ab(n - 2, m - 2);
    if (c <= 0) {
        // a's cursor reaches its last record first (or both together): b stands on its first tag > a[n-2]
        i_exit = n - 1;
        if (c == 0) j_exit = m - 1;
        else if (match[n - 2] != kNone) j_exit = size_t(match[n - 2]) + 1;
        else {
            size_t lo = 0, hi = m;                           // first b tag greater than a[n-2]
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cmp_ab(n - 2, mid) >= 0) lo = mid + 1; else hi = mid; }
            j_exit = lo;
        }
        for (size_t k = 0; k + 1 < n; ++k)
            if (match[k] != kNone) pairs.emplace_back(oa[k], ob[match[k]]);
    } else {
        j_exit = m - 1;
        size_t lo = 0, hi = n;                               // first a tag greater than b[m-2]
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cmp_ab(mid, m - 2) <= 0) lo = mid + 1; else hi = mid; }
        i_exit = lo;
        for (size_t k = 0; k < i_exit; ++k)
            if (match[k] != kNone) pairs.emplace_back(oa[k], ob[match[k]]);
    }
    const size_t before = pairs.size();
    const bool last_equal = cmp_ab(i_exit, j_exit) == 0;     // hpp:317-340 "check 2 last records"
    if (last_equal) pairs.emplace_back(oa[i_exit], ob[j_exit]);
    unmatched = i_exit + j_exit - 2 * before + (last_equal ? 0 : 1);
}

} // namespace fqdhost
