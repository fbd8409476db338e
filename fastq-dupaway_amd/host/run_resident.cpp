// run_resident.cpp — the runs whose text is resident in HBM: files to the device as they lie on disk (BGZF members
// inflated there, records cut there), ordered (run_ordered_resident) and `--unordered` (run_unordered_resident).
#include "run_common.hpp"

namespace fqdhost {
using namespace detail;

namespace detail {

bool inflate_on_device()
{
    const char* v = std::getenv("FQD_GUNZIP_DEVICE");
    return !v || std::atoi(v) != 0;
}

// false: not such a file (nothing is reported; the caller reads it the host way).
// `into` given: the members are inflated into into->text batch by batch WHILE the file is still being read (a batch =
// a few rounds of the chip's waves, one member each: fqd_bgzf_inflate_async on a small engine of this thread's own), and
// the room for the text — sized from the file's size before anything is known about its members, regrown if that was
// too little — is allocated by a helper thread under the first reads: on a device whose free memory another process
// has just given back, hipMalloc clears tens of gigabytes of pages and takes seconds (0.36 - 3.07 s of
// configs[4]'s wall).
bool fetch_bgzf(const std::string& name, size_t block_bytes, int device, CompressedOnDevice& c, FileOnDevice* into)
{
    uint64_t size = 0;
    if (!has_gz_extension(name) || !is_regular_file(name, size) || size < 28) return false;
    {
        // the first member's header decides before anything is allocated: an ordinary gzip file (fetch_gzip_ordinary's) used to cost
        // this function two pinned blocks, an engine, room for six times its size and the read of its first block — 0.15 s of a 1.2 s run
        unsigned char head[18];
        std::FILE* peek = std::fopen(name.c_str(), "rb");
        const bool got = peek && std::fread(head, 1, sizeof head, peek) == sizeof head;
        if (peek) std::fclose(peek);
        size_t data_off = 0;
        if (!got || bgzf_member_size(head, sizeof head, &data_off) == 0) return false;
    }
    InputFile file(name, true);
    HIP_OK(hipSetDevice(device));
    hipStream_t up = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
    // two blocks: the copy of one to HBM runs under the read of the next (its event is waited for before the block is read into again)
    Pinned<char> blocks[2];
    hipEvent_t sent[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    struct SentGuard { hipEvent_t* e; ~SentGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } sent_guard{sent};
    for (int k = 0; k < 2; ++k) { blocks[k].reserve(block_bytes); HIP_OK(hipEventCreateWithFlags(&sent[k], hipEventDisableTiming)); }
    c.bytes.reserve(size + 64);
    // ---- inflate under the read ---------------------------------------------------------------------------------
    static const bool overlap = [] { const char* v = std::getenv("FQD_INFLATE_OVERLAP"); return !v || std::atoi(v) != 0; }();
    const bool inflating = into != nullptr && overlap;
    constexpr uint64_t kBatchMembers = 32768;                 // eight rounds of the chip's 4096 waves, one member each: 2 GB of text, ~10 ms
    std::unique_ptr<EngineHandle> codec;                      // this thread's engine: scratch and stream of the inflate launches
    hipStream_t codec_stream = nullptr;
    struct CodecGuard { hipStream_t& s; std::unique_ptr<EngineHandle>& e; ~CodecGuard() { e.reset(); if (s) (void)hipStreamDestroy(s); } } cg{codec_stream, codec};
    std::thread room;                                          // allocates into->text
    std::exception_ptr room_error;
    struct RoomGuard { std::thread& t; ~RoomGuard() { if (t.joinable()) t.join(); } } rg{room};
    struct Batch { Device<uint64_t> comp_off, out_off; Device<uint32_t> comp_len, out_len, crc; };
    std::vector<std::unique_ptr<Batch>> batches;              // alive until the stream has drained
    Device<uint64_t> d_bad;
    uint64_t launched = 0, launched_bytes = 0;                // members / text bytes handed to the device so far
    hipEvent_t copied = nullptr;
    struct EventGuard { hipEvent_t& e; ~EventGuard() { if (e) (void)hipEventDestroy(e); } } eg{copied};
    if (inflating) {
        HIP_OK(hipStreamCreateWithFlags(&codec_stream, hipStreamNonBlocking));
        codec = std::make_unique<EngineHandle>(1, device, codec_stream);
        HIP_OK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        d_bad.reserve(2);
        HIP_OK(hipMemsetAsync(d_bad.p, 0, 2 * sizeof(uint64_t), codec_stream));
        const uint64_t guess = size * 6 + (64u << 20);        // level-1 FASTQ inflates 4-5.6x
        room = std::thread([&, guess] {
            try { HIP_OK(hipSetDevice(device)); StageClock::Scope t("  on the GPU: room for the text (under the read)"); into->text.room_for(guess, nullptr); }
            catch (...) { room_error = std::current_exception(); }
        });
    }
    auto launch_batch = [&](bool last) {
        const uint64_t have = c.comp_off.size();
        if (!inflating || have == launched || (!last && have - launched < kBatchMembers)) return;
        if (room.joinable()) { room.join(); if (room_error) std::rethrow_exception(room_error); }
        const uint64_t n = have - launched, need = c.text_bytes + 64;
        if (need > into->text.cap) {                           // the guess was too small: everything inflated so far moves
            HIP_OK(hipStreamSynchronize(codec_stream));
            into->text.used = launched_bytes;
            into->text.room_for(std::max<uint64_t>(need, into->text.cap + into->text.cap / 2) - into->text.used, codec_stream);
        }
        batches.emplace_back(new Batch());
        Batch& b = *batches.back();
        b.comp_off.reserve(n); b.out_off.reserve(n); b.comp_len.reserve(n); b.out_len.reserve(n); b.crc.reserve(n);
        // the member arrays go up on the COPY stream: waiting for them must not wait for the batch before this one
        HIP_OK(hipMemcpyAsync(b.comp_off.p, c.comp_off.data() + launched, n * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.out_off.p, c.out_off.data() + launched, n * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.comp_len.p, c.comp_len.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.out_len.p, c.out_len.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipMemcpyAsync(b.crc.p, c.crc.data() + launched, n * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        HIP_OK(hipStreamSynchronize(up));                      // (the vectors may grow and move under the next block's walk)
        HIP_OK(hipEventRecord(copied, up));                    // arrays and compressed bytes of these members are on the device
        HIP_OK(hipStreamWaitEvent(codec_stream, copied, 0));
        if (fqd_bgzf_inflate_async(codec->e, reinterpret_cast<const uint8_t*>(c.bytes.p), b.comp_off.p, b.comp_len.p, b.out_off.p, b.out_len.p,
                                   b.crc.p, n, reinterpret_cast<uint8_t*>(into->text.p), d_bad.p) != FQD_OK)
            throw DeviceError(std::string("GPU engine: ") + fqd_last_error(codec->e));
        launched = have; launched_bytes = c.text_bytes;
    };
    std::string tail;                          // bytes already read from `tail_at` on: a member may straddle two blocks
    uint64_t tail_at = 0, at = 0, member = 0;  // file offsets: of the tail, of the current block, of the member being parsed
    for (int turn = 0;; turn ^= 1) {
        Pinned<char>& block = blocks[turn];
        if (in_flight[turn]) { HIP_OK(hipEventSynchronize(sent[turn])); in_flight[turn] = false; }
        const size_t got = file.read(block.p, block_bytes, host_threads());
        if (got == 0) break;
        if (at + got > size) { (void)hipStreamSynchronize(up); return false; }     // the file grew under us
        HIP_OK(hipMemcpyAsync(c.bytes.p + at, block.p, got, hipMemcpyHostToDevice, up));
        HIP_OK(hipEventRecord(sent[turn], up)); in_flight[turn] = true;
        auto fetch = [&](uint64_t from, size_t len, unsigned char* dst) {
            if (from + len > at + got) return false;
            for (size_t k = 0; k < len; ++k)
                dst[k] = static_cast<unsigned char>(from + k >= at ? block.p[from + k - at] : tail[from + k - tail_at]);
            return true;
        };
        bool ok = true;
        for (;;) {
            unsigned char head[18], trailer[8];
            if (!fetch(member, sizeof head, head)) break;
            size_t data_off = 0;
            const size_t total = bgzf_member_size(head, sizeof head, &data_off);
            if (total == 0) { ok = false; break; }                     // not BGZF (or an extra field of another shape)
            if (!fetch(member + total - 8, sizeof trailer, trailer)) break;
            const uint32_t crc = trailer[0] | (uint32_t(trailer[1]) << 8) | (uint32_t(trailer[2]) << 16) | (uint32_t(trailer[3]) << 24);
            const uint32_t isize = trailer[4] | (uint32_t(trailer[5]) << 8) | (uint32_t(trailer[6]) << 16) | (uint32_t(trailer[7]) << 24);
            if (isize > 65536u) { ok = false; break; }
            if (isize) {
                c.comp_off.push_back(member + data_off); c.comp_len.push_back(static_cast<uint32_t>(total - data_off - 8));
                c.out_off.push_back(c.text_bytes); c.out_len.push_back(isize); c.crc.push_back(crc);
                c.text_bytes += isize;
            }
            member += total;
        }
        if (!ok) { (void)hipStreamSynchronize(up); if (inflating) (void)hipStreamSynchronize(codec_stream); return false; }
        launch_batch(false);
        std::string keep;
        if (member < at + got) {
            if (member < at) keep.assign(tail, static_cast<size_t>(member - tail_at), std::string::npos);
            const uint64_t from = std::max(member, at);
            keep.append(block.p + (from - at), static_cast<size_t>(at + got - from));
        }
        tail.swap(keep); tail_at = member;
        at += got;
    }
    HIP_OK(hipStreamSynchronize(up));                          // every byte of the file is in HBM
    const bool whole = at == size && member == size && c.text_bytes > 0;
    if (inflating) {
        if (whole) launch_batch(true);
        HIP_OK(hipStreamSynchronize(codec_stream));
        if (room.joinable()) { room.join(); if (room_error) std::rethrow_exception(room_error); }
        if (whole) {
            uint64_t bad[2] = {0, 0};
            HIP_OK(hipMemcpy(bad, d_bad.p, sizeof bad, hipMemcpyDeviceToHost));
            c.bad_members = bad[0] + bad[1];
            c.inflated = true;
            StageClock::Scope t("  on the GPU: compressed bytes freed");
            c.bytes.release();
        }
    }
    return whole;
}

// The GPU's share of a file that arrived compressed: inflate, count lines, cut into records.  false: a damaged
// member, or text that is not whole records — the caller reads the file the host way, which says what is wrong.
bool finish_on_device(fqd_engine* e, hipStream_t stream, Format format, CompressedOnDevice& c, FileOnDevice& f)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    if (c.inflated) {                                         // fetch_bgzf did it under the read
        if (c.bad_members) return false;
        return records_on_device(e, stream, format, c.text_bytes, f);
    }
    const uint64_t members = c.comp_off.size();
    Device<uint64_t> d_comp_off, d_out_off; Device<uint32_t> d_comp_len, d_out_len, d_crc;
    d_comp_off.reserve(members); d_out_off.reserve(members); d_comp_len.reserve(members); d_out_len.reserve(members); d_crc.reserve(members);
    HIP_OK(hipMemcpy(d_comp_off.p, c.comp_off.data(), members * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_out_off.p, c.out_off.data(), members * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_comp_len.p, c.comp_len.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_out_len.p, c.out_len.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_crc.p, c.crc.data(), members * sizeof(uint32_t), hipMemcpyHostToDevice));
    { StageClock::Scope t("  on the GPU: room for the text"); f.text.room_for(c.text_bytes + 64, stream); }
    uint64_t bad = 0;
    {
        StageClock::Scope t("  on the GPU: inflate + CRC check");
        engine_ok(fqd_bgzf_inflate(e, reinterpret_cast<const uint8_t*>(c.bytes.p), d_comp_off.p, d_comp_len.p, d_out_off.p, d_out_len.p,
                                   d_crc.p, members, reinterpret_cast<uint8_t*>(f.text.p), &bad));
    }
    { StageClock::Scope t("  on the GPU: compressed bytes freed"); c.bytes.release(); }
    if (bad) return false;
    return records_on_device(e, stream, format, c.text_bytes, f);
}

// The text of a file is in HBM (f.text, text_bytes of it): cut it into records there.  false: not whole records.
bool records_on_device(fqd_engine* e, hipStream_t stream, Format format, uint64_t text_bytes, FileOnDevice& f)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    struct { uint64_t text_bytes; } c{text_bytes};
    const uint32_t lines_per_record = format == Format::Fastq ? 4u : 2u;
    uint64_t lines = 0;
    { StageClock::Scope t("  on the GPU: line count"); engine_ok(fqd_count_lines(e, reinterpret_cast<const uint8_t*>(f.text.p), c.text_bytes, &lines)); }
    const uint64_t n = lines / lines_per_record;
    {
        StageClock::Scope t("  on the GPU: room for the record arrays");
        f.start.room_for(n, stream); f.seq_off.room_for(n, stream); f.id_len.room_for(n, stream); f.seq_len.room_for(n, stream); f.size.room_for(n, stream);
    }
    int well_formed = 0;
    {
        StageClock::Scope t("  on the GPU: record scan");
        engine_ok(fqd_scan_records(e, reinterpret_cast<const uint8_t*>(f.text.p), c.text_bytes, lines_per_record, n,
                                   f.start.p, f.seq_off.p, f.id_len.p, f.seq_len.p, f.size.p, &well_formed));
    }
    if (!well_formed || n == 0) return false;
    f.text.used = c.text_bytes;
    f.start.used = f.seq_off.used = f.id_len.used = f.seq_len.used = f.size.used = n;
    f.n = n;
    return true;
}

// What the dedup engine of a resident run will hold, guessed from the sizes of the input files before anything of them has
// been read, so that its key store and table can be allocated — and their pages cleared by the driver — on a helper thread
// under the reads (the key store's first hipMalloc was 1.05 s of a 1.13 s "pair dedup" stage).  A guess
// that is too small costs what it always cost (the store grows); one too large costs HBM nobody else wants.
void guess_capacity(int S, const std::string* in, uint64_t& reads, uint64_t& bases)
{
    reads = bases = 0;
    uint64_t text[2] = {0, 0};
    for (int s = 0; s < S; ++s) {
        uint64_t size = 0;
        if (!is_regular_file(in[s], size)) { reads = bases = 0; return; }
        text[s] = has_gz_extension(in[s]) ? size * 5 : size;
    }
    const uint64_t least = S == 2 ? std::min(text[0], text[1]) : text[0];
    reads = least / 280 + 1024;                                // a 150-base FASTQ record is ~316 bytes
    bases = (text[0] + text[1]) / 2 + 4096;                    // about half of FASTQ text is sequence
}

// A plain regular file as it is to the tail of f.text (a pinned block, parallel preads, H2D); false: not such a file.
bool fetch_plain(const std::string& name, size_t block_bytes, int device, FileOnDevice& f, uint64_t& text_bytes)
{
    uint64_t size = 0;
    if (has_gz_extension(name) || !is_regular_file(name, size) || size == 0) return false;
    InputFile file(name, true);
    HIP_OK(hipSetDevice(device));
    hipStream_t up = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
    Pinned<char> block[2];
    block[0].reserve(block_bytes); block[1].reserve(block_bytes);
    f.text.room_for(size + 64, up);
    uint64_t at = 0;
    for (int k = 0;; k ^= 1) {                                   // the copy of one block overlaps the read of the next
        const size_t got = file.read(block[k].p, block_bytes, host_threads());
        HIP_OK(hipStreamSynchronize(up));                          // the other block's copy
        if (got == 0) break;
        if (at + got > size) return false;                         // the file grew under us
        HIP_OK(hipMemcpyAsync(f.text.p + at, block[k].p, got, hipMemcpyHostToDevice, up));
        at += got;
    }
    text_bytes = at;
    return at == size;
}

// An ORDINARY gzip file (members that are one long deflate stream each: what gzip, pigz and sequencers write; the reference reads it
// through the same decompressor as any .gz, file_utils.cpp:59-66) to HBM as it lies on disk and inflated THERE (fqd_gunzip:
// block starts guessed per unit, every unit decoded into two texts behind made-up windows, windows chained, bytes): 25 GB/s of text on one
// MI355X against 5.4 for the several-thread host reader (host/pgzip.hpp), which stays the way for pipes, small files and
// whatever is irregular.  FQD_GUNZIP_ORDINARY_DEVICE=0 turns it off.
// false: not such a file, or anything irregular (a guess that did not hold and could not be repaired, damage, CRC or length
// other than a trailer's, bytes behind the last member): nothing is reported, the caller reads the file the host way.
bool fetch_gzip_ordinary(const std::string& name, size_t block_bytes, int device, FileOnDevice& f, uint64_t& text_bytes)
{
    static const bool wanted = [] { const char* v = std::getenv("FQD_GUNZIP_ORDINARY_DEVICE"); return !v || std::atoi(v) != 0; }();
    uint64_t size = 0;
    if (!wanted || !has_gz_extension(name) || !is_regular_file(name, size) || size < 28) return false;
    // the member's header (RFC 1952): magic, method 8, the optional fields; a BGZF file never comes here (fetch_bgzf took it)
    size_t header = 0;
    {
        unsigned char p[1024];
        std::FILE* peek = std::fopen(name.c_str(), "rb");
        const size_t got = peek ? std::fread(p, 1, sizeof p, peek) : 0;
        if (peek) std::fclose(peek);
        if (got < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || (p[3] & 0xE0)) return false;
        size_t h = 10;
        if (p[3] & 4) { if (h + 2 > got) return false; h += 2 + (p[h] | (size_t(p[h + 1]) << 8)); }
        if (p[3] & 8) { while (h < got && p[h]) ++h; ++h; }
        if (p[3] & 16) { while (h < got && p[h]) ++h; ++h; }
        if (p[3] & 2) h += 2;
        if (h + 10 > got || h + 10 > size) return false;
        header = h;
    }
    InputFile file(name, true);
    HIP_OK(hipSetDevice(device));
    hipStream_t up = nullptr, work = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    struct Guard { hipStream_t& s; ~Guard() { if (s) (void)hipStreamDestroy(s); } } g{up}, gw{work};
    HIP_OK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
    Device<char>& comp = f.packed;                                 // (stays with the file: see FileOnDevice)
    comp.reserve(size + 64);
    HIP_OK(hipMemsetAsync(comp.p + size, 0, 64, up));
    HIP_OK(hipStreamSynchronize(up));
    // The inflating runs on a thread of its own WHILE the file is read (fqd_gunzip_arriving): block starts are looked for and units
    // decoded as their bytes reach HBM; `arrived` is raised here after every block's copy.  Room for the text: ISIZE is its length
    // modulo 2^32 only and stands at the file's end, so room is what FASTQ at its most packable needs (binned qualities: sevenfold);
    // a text that outgrows it sends the file to the host reader.
    const uint64_t room = size * 8u + (64u << 10);
    alignas(8) volatile uint64_t arrived = 0;
    struct Outcome { int rc = FQD_OK; uint64_t tb = 0, db = 0; uint32_t crc = 0; int32_t ok = 0; std::string error; std::exception_ptr thrown; } out;
    std::thread worker([&] {
        try {
            HIP_OK(hipSetDevice(device));
            { StageClock::Scope t("  ordinary gzip: room for the text (under the read)"); f.text.room_for(room + 64, work); }
            f.codec = std::make_shared<EngineHandle>(1, device, nullptr);   // a small engine of this file's own: stream, scratch and error slot of the call
            EngineHandle& eng = *f.codec;
            StageClock::Scope t("  ordinary gzip: inflated on the GPU (fqd_gunzip_arriving, under the read and after it)");
            out.rc = fqd_gunzip_arriving(eng.e, reinterpret_cast<const uint8_t*>(comp.p) + header, size - header, &arrived,
                                         reinterpret_cast<uint8_t*>(f.text.p), room, &out.tb, &out.db, &out.crc, &out.ok);
            if (out.rc != FQD_OK) out.error = fqd_last_error(eng.e);
        } catch (...) { out.thrown = std::current_exception(); }
    });
    struct Join { std::thread& t; volatile uint64_t& arrived; ~Join() { if (t.joinable()) { arrived = ~0ull; t.join(); } } } join{worker, arrived};   // (an early way out: "the rest will not come")
    Pinned<char> block[2];
    block[0].reserve(block_bytes); block[1].reserve(block_bytes);
    uint64_t at = 0, sent[2] = {0, 0};
    {
        StageClock::Scope reading("  ordinary gzip: the file read and copied to HBM");
        for (int k = 0;; k ^= 1) {                                   // the copy of one block overlaps the read of the next
            const size_t got = file.read(block[k].p, block_bytes, host_threads());
            HIP_OK(hipStreamSynchronize(up));                          // the other block's copy: those bytes are in HBM now
            if (sent[k ^ 1] > header) arrived = sent[k ^ 1] - header;
            if (got == 0) break;
            if (at + got > size) return false;                         // the file grew under us
            HIP_OK(hipMemcpyAsync(comp.p + at, block[k].p, got, hipMemcpyHostToDevice, up));
            at += got;
            sent[k] = at;
        }
    }
    if (at != size) return false;
    arrived = size - header;
    { StageClock::Scope t("  ordinary gzip: the rest of the inflating, after the read"); worker.join(); }
    if (out.thrown) std::rethrow_exception(out.thrown);
    if (out.rc != FQD_OK) throw DeviceError(std::string("GPU engine: ") + out.error);
    // (every member's CRC-32 and ISIZE were held against its trailer by the call)
    static const bool park = [] { const char* v = std::getenv("FQD_GUNZIP_PARK"); return !v || std::atoi(v) != 0; }();   // 0: give the scratch and the packed bytes back at once (A/B)
    if (!park) { f.codec.reset(); f.packed.release(); }
    if (!out.ok || header + out.db + 8 != size) { f.text.used = 0; return false; }
    text_bytes = out.tb;
    return true;
}

} // namespace detail

// An ordered run (single-end, or paired files read side by side) with a codec at either end — BGZF inputs, or `.gz`
// outputs of plain regular inputs: the files go to HBM as they lie on disk, are inflated (if compressed) and cut into
// records there, every read (pair) is deduplicated where it lies, and the
// survivors leave in input order window by window (deflated on the device for `.gz` outputs).  Taken only when
// everything is plain sailing — regular BGZF files of whole records, as many in file 2 as in file 1, no unknown
// base, everything fits in HBM; otherwise false is returned BEFORE any output is touched and the streaming run
// (run_ordered), which reproduces the reference's behaviour for every irregular input, does the job.
bool HashDupRemover::run_ordered_resident(int S, const std::string* in, const std::string* out)
{
    if (const char* v = std::getenv("FQD_ORDERED_RESIDENT")) if (std::atoi(v) == 0) return false;
    if (!inflate_on_device()) return false;
    // worth it when a codec is involved: a BGZF input, or a `.gz` output the GPU can deflate (plain files in and
    // out are better off in the streaming run, where reading, the GPU and writing overlap)
    bool any_gz_in = false, any_gz_out = false;
    for (int s = 0; s < S; ++s) {
        uint64_t size = 0;
        if (!is_regular_file(in[s], size)) return false;
        any_gz_in |= has_gz_extension(in[s]);
        any_gz_out |= has_gz_extension(out[s]);
    }
    if (!any_gz_in && !(any_gz_out && deflate_on_device())) return false;
    HIP_OK(hipSetDevice(tuning_.device));
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));
    // a file fetched whole is read in larger pieces than the streaming run's blocks: the parallel read of a piece needs 16 MB per thread
    const size_t fetch_bytes = std::max<size_t>(block_bytes, std::min<size_t>(64u << 20, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : (64 << 20))));
    FileOnDevice dev[2];
    Device<uint8_t> keep;
    uint64_t n = 0, dups = 0;
    std::unique_ptr<EngineHandle> eng;
    SurvivorBuffers buffers;
    std::thread make_engine; std::exception_ptr engine_error;
    struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } join_guard{make_engine};
    try {
        CompressedOnDevice packed[2];
        bool fetched[2] = {false, false};
        bool text_ready[2] = {false, false};                       // a `.gz` input whose TEXT is in HBM already (fetch_gzip_ordinary)
        std::exception_ptr fetch_error[2];
        uint64_t plain_bytes[2] = {0, 0};
        // the engine — its key store and table sized from the files' sizes — is made on a helper thread under the reads
        uint64_t cap_reads = 0, cap_bases = 0;
        guess_capacity(S, in, cap_reads, cap_bases);
        make_engine = std::thread([&] {
            try { HIP_OK(hipSetDevice(tuning_.device)); StageClock::Scope t("  on the GPU: engine, key store, table (under the read)"); eng = std::make_unique<EngineHandle>(S, tuning_.device, stream, cap_reads, cap_bases); }
            catch (...) { engine_error = std::current_exception(); }
        });
        {
            StageClock::Scope t("ordered/resident: files to HBM");
            auto fetch = [&](int s) {
                try {
                    fetched[s] = has_gz_extension(in[s]) ? fetch_bgzf(in[s], fetch_bytes, tuning_.device, packed[s], &dev[s])
                                                         : fetch_plain(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                    if (!fetched[s] && has_gz_extension(in[s])) {              // not BGZF: an ordinary gzip file, inflated on the GPU too if asked for
                        packed[s] = CompressedOnDevice(); dev[s].forget();
                        fetched[s] = text_ready[s] = fetch_gzip_ordinary(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                    }
                } catch (const DeviceOutOfMemory&) { fetched[s] = false; }
                catch (const DeviceError&) { fetched[s] = false; fetch_error[s] = std::current_exception(); }
                catch (const std::exception&) { fetched[s] = false; }      // the host reader will say what is wrong with the file
            };
            std::thread second;
            if (S == 2) second = std::thread(fetch, 1);
            fetch(0);
            if (S == 2) second.join();
        }
        if (make_engine.joinable()) make_engine.join();
        for (int s = 0; s < S; ++s) if (fetch_error[s]) std::rethrow_exception(fetch_error[s]);
        for (int s = 0; s < S; ++s) if (!fetched[s]) return false;
        if (engine_error) std::rethrow_exception(engine_error);
        {
            StageClock::Scope t("ordered/resident: inflate + record scan on the GPU");
            for (int s = 0; s < S; ++s) {
                const bool ok = has_gz_extension(in[s]) && !text_ready[s] ? finish_on_device(eng->e, stream, format_, packed[s], dev[s])
                                                                          : records_on_device(eng->e, stream, format_, plain_bytes[s], dev[s]);
                if (!ok) return false;
            }
        }
        if (S == 2 && dev[0].n != dev[1].n) return false;
        n = dev[0].n;
        StageClock::Scope t("ordered/resident: dedup on the GPU");
        keep.reserve(n);
        const size_t kBatch = 16u << 20;
        int rc = FQD_OK;
        for (size_t a = 0; a < n && rc == FQD_OK; a += kBatch) {
            fqd_reads seg[2] = {};
            for (int s = 0; s < S; ++s) {
                seg[s].bases = reinterpret_cast<const uint8_t*>(dev[s].text.p);
                seg[s].offsets = dev[s].seq_off.p + a; seg[s].lengths = dev[s].seq_len.p + a;
            }
            rc = (a + kBatch < n ? fqd_submit : fqd_submit_final)(eng->e, seg, std::min<size_t>(kBatch, n - a), FQD_MEM_DEVICE, keep.p + a);
        }
        if (rc == FQD_OK) rc = fqd_engine_sync(eng->e);
        if (rc == FQD_ERR_BAD_BASE) return false;                 // the streaming run cuts the output where the reference does
        if (rc != FQD_OK) throw DeviceError(std::string("GPU engine: ") + fqd_last_error(eng->e));
        if (std::getenv("FQD_TEST_FAIL_RESIDENT")) throw DeviceError("GPU engine: forced by FQD_TEST_FAIL_RESIDENT");      // tests: the hand-over is announced
        fqd_stats st{};
        fqd_get_stats(eng->e, &st);
        dups = st.duplicates;
        // everything the writer needs is reserved HERE, while the run can still hand over: once an output exists it cannot
        bool gz_out[2] = {false, false};
        for (int s = 0; s < S; ++s) gz_out[s] = has_gz_extension(out[s]);
        FileOnDevice* files[2] = {&dev[0], &dev[1]};
        const uint32_t* idx[2] = {nullptr, nullptr};
        plan_survivors(eng->e, S, files, idx, keep.p, n, gz_out, memlimit_, buffers);
    } catch (const DeviceOutOfMemory&) {
        return false;                                             // HBM that does not suffice: the streaming run needs a few blocks of it only
    } catch (const DeviceError& e) {
        announce_handover("the GPU-resident ordered run", e);     // nothing has been written yet
        return false;
    } catch (const std::exception&) {
        return false;                                             // an input the host reader will report on in the reference's words
    }
    // from here on the run is this one's: outputs are created, filled and closed
    OutputFile sink0(out[0]);
    std::unique_ptr<OutputFile> sink1;
    if (S == 2) sink1 = std::make_unique<OutputFile>(out[1]);
    OutputFile* sinks[2] = {&sink0, sink1.get()};
    {
        StageClock::Scope t("ordered/resident: survivors out of HBM");
        FileOnDevice* files[2] = {&dev[0], &dev[1]};
        const uint32_t* idx[2] = {nullptr, nullptr};
        write_survivors(eng->e, stream, S, files, idx, keep.p, n, dups, sinks, format_, memlimit_, true, &buffers);
    }
    if (tuning_.leave_memory_to_exit) g_leave_memory_to_exit = true;
    StageClock::report();
    summary_.total = n; summary_.duplicates = dups; summary_.unmatched = 0;
    if (verbose_) {
        if (S == 1) std::cout << summary_.total << " reads processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        else        std::cout << summary_.total << " read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
    }
    return true;
}

void HashDupRemover::run_unordered_resident(const std::string* in, const std::string* out)
{
    HIP_OK(hipSetDevice(tuning_.device));
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    // the engine — key store and table sized from the files' sizes — is made on a helper thread under the reads of the
    // inputs: its first use comes after them (guess_capacity)
    JoinedPairs jp;                                           // (before `eng`: the helper thread makes room in it, and eng's destructor joins that thread first)
    struct LazyEngine {
        std::unique_ptr<EngineHandle> holder; std::thread maker; std::exception_ptr error;
        ~LazyEngine() { if (maker.joinable()) maker.join(); }
        fqd_engine* get() { if (maker.joinable()) maker.join(); if (error) std::rethrow_exception(error); return holder->e; }
    } eng;
    {
        uint64_t cap_reads = 0, cap_bases = 0;
        guess_capacity(2, in, cap_reads, cap_bases);
        eng.maker = std::thread([this, &eng, &jp, stream, cap_reads, cap_bases] {
            try {
                HIP_OK(hipSetDevice(tuning_.device));
                StageClock::Scope t("  on the GPU: engine, key store, table, join arrays (under the read)");
                eng.holder = std::make_unique<EngineHandle>(2, tuning_.device, stream, cap_reads, cap_bases);
                if (cap_reads) jp.prepare(cap_reads);
            }
            catch (...) { eng.error = std::current_exception(); }
        });
    }
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(eng.get())); };
    const size_t block_bytes = std::max<size_t>(1u << 20, std::min<size_t>(tuning_.block_bytes, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : tuning_.block_bytes)));
    // a file fetched whole is read in larger pieces than the streaming run's blocks: the parallel read of a piece needs 16 MB per thread
    const size_t fetch_bytes = std::max<size_t>(block_bytes, std::min<size_t>(64u << 20, static_cast<size_t>(memlimit_ > 0 ? memlimit_ / 16 : (64 << 20))));

    FileOnDevice dev[2];

    // ---- the one pass: every block to the tail of the file's text in HBM -------------------------------
    {
        StageClock::Scope t("unordered/resident: read, scan, text to HBM");
        // both files at the same time, each on its own thread and copy stream; what goes wrong is still
        // reported in the reference's order: everything about file 1 before anything about file 2 (hpp:161-173)
        std::exception_ptr err[2];
        ParseFailure parse_failure[2];
        auto load = [&](int s) {
            try {
                HIP_OK(hipSetDevice(tuning_.device));
                hipStream_t up = nullptr;
                HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
                struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } g{up};
                Pinned<uint64_t> h_start, h_seq; Pinned<uint32_t> h_idl, h_sql, h_size;
                FileOnDevice& f = dev[s];
                uint64_t known = 0;
                if (is_regular_file(in[s], known) && !has_gz_extension(in[s])) f.text.room_for(known + 64, up);   // no regrowth for plain files
                Side side;
                side.open_file(in[s], format_, true, block_bytes);
                side.prime(3, tuning_.device);
                while (side.available() > 0) {
                    PooledBlock* b = side.cur;
                    const size_t from = side.pos, nb = b->recs.size() - from;
                    const RecordRef* r = &b->recs[from];
                    const uint64_t text_lo = r[0].start, bytes = r[nb - 1].start + r[nb - 1].size - text_lo;
                    f.text.room_for(bytes + 64, up);
                    HIP_OK(hipMemcpyAsync(f.text.p + f.text.used, b->text.p + text_lo, bytes, hipMemcpyHostToDevice, up));
                    h_start.reserve(nb); h_seq.reserve(nb); h_idl.reserve(nb); h_sql.reserve(nb); h_size.reserve(nb);
                    for (size_t k = 0; k < nb; ++k) {
                        h_start.p[k] = f.text.used + (r[k].start - text_lo); h_seq.p[k] = h_start.p[k] + r[k].id_len;
                        h_idl.p[k] = r[k].id_len; h_sql.p[k] = r[k].seq_len; h_size.p[k] = r[k].size;
                    }
                    f.start.room_for(nb, up); f.seq_off.room_for(nb, up); f.id_len.room_for(nb, up); f.seq_len.room_for(nb, up); f.size.room_for(nb, up);
                    HIP_OK(hipMemcpyAsync(f.start.p + f.n, h_start.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.seq_off.p + f.n, h_seq.p, nb * sizeof(uint64_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.id_len.p + f.n, h_idl.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.seq_len.p + f.n, h_sql.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipMemcpyAsync(f.size.p + f.n, h_size.p, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up));
                    HIP_OK(hipStreamSynchronize(up));            // the block and the staging arrays are reused
                    f.text.used += bytes;
                    f.start.used = f.seq_off.used = f.id_len.used = f.seq_len.used = f.size.used = f.n + nb;
                    f.n += nb;
                    side.pos += nb;
                }
                if (side.failed) parse_failure[s] = side.failure;
            } catch (...) { err[s] = std::current_exception(); }
        };
        CompressedOnDevice packed[2];
        bool on_device[2] = {false, false};
        bool plain_on_device[2] = {false, false};              // a plain regular file: copied to HBM as it is, cut into records there
        uint64_t plain_bytes[2] = {0, 0};
        auto fetch_or_load = [&](int s) {
            if (inflate_on_device()) {
                try {
                    if (has_gz_extension(in[s])) {
                        on_device[s] = fetch_bgzf(in[s], fetch_bytes, tuning_.device, packed[s], &dev[s]);
                        if (!on_device[s]) {                                   // not BGZF: an ordinary gzip file, inflated on the GPU too if asked for
                            packed[s] = CompressedOnDevice(); dev[s].forget();
                            plain_on_device[s] = fetch_gzip_ordinary(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                        }
                    }
                    else plain_on_device[s] = fetch_plain(in[s], fetch_bytes, tuning_.device, dev[s], plain_bytes[s]);
                }
                catch (const DeviceOutOfMemory&) { err[s] = std::current_exception(); return; }   // rethrown below: the two-pass run takes over
                catch (const std::exception&) { on_device[s] = plain_on_device[s] = false; }          // the host way will say what is wrong
            }
            if (!on_device[s] && !plain_on_device[s]) { packed[s] = CompressedOnDevice(); dev[s].forget(); load(s); }
        };
        std::thread second(fetch_or_load, 1);
        fetch_or_load(0);
        second.join();
        for (int s = 0; s < 2; ++s) {
            if (!on_device[s] && !plain_on_device[s]) continue;
            StageClock::Scope t2("unordered/resident: inflate + record scan on the GPU");
            const bool ok = on_device[s] ? finish_on_device(eng.get(), stream, format_, packed[s], dev[s])
                                         : records_on_device(eng.get(), stream, format_, plain_bytes[s], dev[s]);
            if (!ok) {                                                                  // read it again the host way: that one reports
                dev[s].forget();
                packed[s] = CompressedOnDevice();
                load(s);
            }
        }
        for (int s = 0; s < 2; ++s) {
            if (err[s]) std::rethrow_exception(err[s]);
            if (parse_failure[s].set) { std::cerr << parse_failure[s].diag; throw std::runtime_error(parse_failure[s].what); }
        }
        for (int s = 0; s < 2; ++s) {
            FileOnDevice& f = dev[s];
            f.tag_off.reserve(f.n); f.tag_len.reserve(f.n);
            engine_ok(fqd_extract_tags(eng.get(), reinterpret_cast<const uint8_t*>(f.text.p), f.start.p, f.id_len.p, f.n, f.tag_off.p, f.tag_len.p));
        }
    }

    DeviceSide side[2];
    for (int s = 0; s < 2; ++s) {
        side[s].tag_bytes = side[s].seq_bytes = reinterpret_cast<const uint8_t*>(dev[s].text.p);
        side[s].tag_off = dev[s].tag_off.p; side[s].tag_len = dev[s].tag_len.p;
        side[s].seq_off = dev[s].seq_off.p; side[s].seq_len = dev[s].seq_len.p; side[s].n = dev[s].n;
    }
    // The join, the dedup and every buffer the writer needs come BEFORE the outputs exist: HBM that does not suffice for
    // them (DeviceOutOfMemory) still hands the job to the two-pass run.  On disk nothing differs from the reference's
    // order — outputs opened after the sort phase, then the merge (hpp:265-266) — a bad base found by the dedup cuts the
    // output at the same pair either way.
    fqd_engine* engine_now = eng.get();                        // (joins the helper thread: jp is ours from here on)
    join_and_dedup(engine_now, stream, side, tuning_.reference_tail_rule, jp);
    const uint64_t n_proc = jp.n_proc, upto = std::min<uint64_t>(n_proc, jp.written_below);
    uint64_t dups = 0;
    {
        std::vector<uint8_t> keep(upto);
        if (upto) HIP_OK(hipMemcpyAsync(keep.data(), jp.keep.p, upto, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (uint64_t k = 0; k < upto; ++k) dups += keep[k] == 0;
    }
    FileOnDevice* files[2] = {&dev[0], &dev[1]};
    const uint32_t* idx[2] = {jp.pair[0].p, jp.pair[1].p};
    SurvivorBuffers buffers;
    {
        const bool gz_out[2] = {has_gz_extension(out[0]), has_gz_extension(out[1])};
        plan_survivors(eng.get(), 2, files, idx, jp.keep.p, upto, gz_out, memlimit_, buffers);
    }

    OutputFile sink0(out[0]), sink1(out[1]);
    OutputFile* sinks[2] = {&sink0, &sink1};

    // ---- outputs: the device assembles windows of survivors in output order, the host writes them -----
    {
        StageClock::Scope t("unordered/resident: survivors out of HBM");
        write_survivors(eng.get(), stream, 2, files, idx, jp.keep.p, upto, dups, sinks, format_, memlimit_, true, &buffers);
    }
    if (tuning_.leave_memory_to_exit) g_leave_memory_to_exit = true;
    StageClock::report();
    if (jp.bad) throw_unknown_base(jp.bad_byte);
    summary_.total = n_proc; summary_.duplicates = dups; summary_.unmatched = jp.unmatched;
    if (verbose_) {
        std::cout << summary_.total << " valid read pairs processed, out of which " << summary_.duplicates << " duplicates were removed.\n";
        std::cout << summary_.unmatched << " Non-matching entries from both files were skipped.\n";
    }
}

} // namespace fqdhost
