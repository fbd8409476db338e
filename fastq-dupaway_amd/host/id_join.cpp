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
        std::unique_ptr<char[]> buf(new char[cap + 16]);
        size_t have = carry.size();
        if (have) std::memcpy(buf.get(), carry.data(), have);
        carry.clear();
        have += file.read(buf.get() + have, cap - have);
        refs.clear();
        const size_t consumed = scan_records(f, true, buf.get(), have, refs, out.failure);
        if (first && refs.empty() && !out.failure.set)
            throw std::runtime_error("Not enough memory to read a single object!");
        first = false;
        for (const RecordRef& r : refs)
            out.recs.push_back(FileRecord{buf.get() + r.start, r.size, r.id_len, r.seq_len, r.tag_off, r.tag_len});
        const bool at_end = file.eof();
        if (!out.failure.set && !at_end) {
            if (refs.empty()) throw std::runtime_error("Not enough memory to read a single object!");
            carry.assign(buf.get() + consumed, buf.get() + have);
        }
        out.chunks.push_back(std::move(buf));
        if (out.failure.set || at_end) break;
    }
}

static std::vector<uint64_t> order_by_tag(const LoadedFile& f)
{
    std::vector<uint64_t> idx(f.recs.size());
    std::iota(idx.begin(), idx.end(), 0);
    // the reference's std::sort over views with operator< = tag compare (external_sort.hpp:105)
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t x, uint64_t y) {
        const FileRecord& a = f.recs[x]; const FileRecord& b = f.recs[y];
        return compare_tags(a.tag(), a.tag_len, b.tag(), b.tag_len) < 0;
    });
    return idx;
}

void join_by_tag(const LoadedFile& a, const LoadedFile& b, bool tail_rule,
                 std::vector<std::pair<uint64_t, uint64_t>>& pairs, uint64_t& unmatched)
{
    pairs.clear(); unmatched = 0;
    const std::vector<uint64_t> oa = order_by_tag(a), ob = order_by_tag(b);
    const size_t n = oa.size(), m = ob.size();
    if (n == 0 || m == 0) return;
    auto cmp = [&](size_t i, size_t j) {
        const FileRecord& l = a.recs[oa[i]]; const FileRecord& r = b.recs[ob[j]];
        return compare_tags(l.tag(), l.tag_len, r.tag(), r.tag_len);
    };
    size_t i = 0, j = 0;
    if (tail_rule) {
        while (i + 1 < n && j + 1 < m) {                     // hpp:281: neither side on its last record
            const int c = cmp(i, j);
            if (c < 0)      { ++i; ++unmatched; }            // hpp:284-287
            else if (c > 0) { ++j; ++unmatched; }            // hpp:288-290
            else            { pairs.emplace_back(oa[i], ob[j]); ++i; ++j; }
        }
        if (cmp(i, j) == 0) pairs.emplace_back(oa[i], ob[j]); // hpp:317-340 "check 2 last records"
        else ++unmatched;
    } else {
        while (i < n && j < m) {
            const int c = cmp(i, j);
            if (c < 0)      { ++i; ++unmatched; }
            else if (c > 0) { ++j; ++unmatched; }
            else            { pairs.emplace_back(oa[i], ob[j]); ++i; ++j; }
        }
        unmatched += (n - i) + (m - j);
    }
}

} // namespace fqdhost
