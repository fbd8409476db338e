// survivor_writer.cpp — survivors written verbatim in input (or tag) order: the writer threads of the streaming runs
// (SurvivorWriters) and the window-by-window writer of the runs whose text is resident in HBM (plan_survivors,
// write_survivors; reference hash_dup_remover.hpp:135-137,240-243,303-306 write record by record).
#include "run_common.hpp"

namespace fqdhost {
using namespace detail;

namespace detail {

unsigned write_threads()
{
    static const unsigned t = [] { const char* v = std::getenv("FQD_WRITE_THREADS"); const int x = v ? std::atoi(v) : 0; return x > 0 ? unsigned(x) : std::min(8u, host_threads()); }();
    return t;
}

void SurvivorWriters::body(int s)
{
    bool failed_already = false;
    std::vector<OutputFile::Piece> pieces;
    for (;;) {
        Work* w;
        { StageClock::Scope t("writer: wait for a batch"); w = queue_[s].pop(); }
        const bool stop = w->stop;
        if (!stop && !failed_already) {
            StageClock::Scope t("writer: write survivors");
            try {
                const Block& b = *w->blk[s];
                pieces.clear();                          // runs of adjacent survivors, written where they lie
                const char* run_from = nullptr; size_t run_len = 0;
                for (size_t k = 0; k < w->n; ++k) {
                    const RecordRef& r = b.recs[w->begin[s] + k];
                    const bool keep = w->keep.p[k] != 0 && w->first_index + k < w->emit_below;
                    if (keep) {
                        const char* p = b.text.p + r.start;
                        if (run_from && run_from + run_len == p) run_len += r.size;
                        else { if (run_len) pieces.push_back({run_from, run_len}); run_from = p; run_len = r.size; }
                    }
                }
                if (run_len) pieces.push_back({run_from, run_len});
                StageClock::Scope t2("writer: copy out");
                sink_[s]->write_pieces(pieces.data(), pieces.size(), write_threads());
            } catch (...) { error_[s] = std::current_exception(); failed_already = true; }
        }
        if (!stop) w->blk[s]->release();
        Channel<Work>* home = w->home ? w->home : recycle_;
        if (w->writers_left.fetch_sub(1) == 1) home->push(w);
        if (stop) break;
    }
}

// `.gz` outputs of the resident run: deflated on the GPU (fqd_bgzf_deflate; the size of zlib level 1-2 at a
// small fraction of its time) unless a level was asked for — FQD_GZ_LEVEL=N means the host codec at level N —
// or FQD_GZ_DEVICE=0/1 says otherwise.
bool deflate_on_device()
{
    if (const char* v = std::getenv("FQD_GZ_DEVICE")) return std::atoi(v) != 0;
    return std::getenv("FQD_GZ_LEVEL") == nullptr;
}

void plan_survivors(fqd_engine* e, int S, FileOnDevice* const* file, const uint32_t* const* idx, const uint8_t* keep, uint64_t upto,
                           const bool* gz_out, long long memlimit, SurvivorBuffers& b)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw DeviceError(std::string("GPU engine: ") + fqd_last_error(e)); };
    b.window = std::max<uint64_t>(4u << 20, static_cast<uint64_t>(memlimit > 0 ? memlimit : (2ll << 30)) / 16);   // bytes per buffer, two per file
    if (const char* v = std::getenv("FQD_STREAM_WINDOW_KB")) { const long kb = std::atol(v); if (kb > 0) b.window = static_cast<uint64_t>(kb) << 10; }
    b.roomy = b.window + b.window / 4;                      // the most a window may hold
    for (int s = 0; s < S; ++s) {
        SurvivorBuffers::PerFile& o = b.f[s];
        o.src_off.reserve(upto); o.dst_off.reserve(upto + 1); o.len.reserve(upto);
        engine_ok(fqd_output_plan(e, keep, idx[s], upto, file[s]->start.p, file[s]->size.p, o.src_off.p, o.len.p, o.dst_off.p, &o.total));
        o.on_device = gz_out[s] && deflate_on_device();
        // every buffer is sized once, for the largest window the writer lets through (a single record larger than that is the
        // one case that grows them later): a window a little larger than all before it must not cost a new pinned allocation
        const uint64_t room = std::min<uint64_t>(b.roomy, std::max<uint64_t>(o.total, 1));
        for (int k = 0; k < 2; ++k) {
            o.d_win[k].reserve(room + 64);
            o.buf[k].reserve((o.on_device ? std::max<uint64_t>(room / 2, 1u << 20) : room) + 64);
            if (o.on_device) o.d_members[k].reserve(fqd_bgzf_bound(room));
        }
    }
    b.planned = true;
}

void write_survivors(fqd_engine* e, hipStream_t stream, int S, FileOnDevice* const* file, const uint32_t* const* idx,
                            const uint8_t* keep, uint64_t upto, uint64_t dups, OutputFile* const* sinks, Format format, long long memlimit,
                            bool close_sinks, SurvivorBuffers* planned)
{
    auto engine_ok = [&](int rc) { if (rc != FQD_OK) throw std::runtime_error(std::string("GPU engine: ") + fqd_last_error(e)); };
    SurvivorBuffers own;
    if (!planned || !planned->planned) {
        bool gz_out[2] = {false, false};
        for (int s = 0; s < S; ++s) gz_out[s] = sinks[s]->is_gz();
        plan_survivors(e, S, file, idx, keep, upto, gz_out, memlimit, own);
        planned = &own;
    }
    const uint64_t window = planned->window, roomy = planned->roomy;
    const uint32_t lines_per_record = format == Format::Fastq ? 4u : 2u;
    struct Out {
        Device<uint64_t>& src_off; Device<uint64_t>& dst_off; Device<uint32_t>& len; uint64_t total;
        Pinned<char>* buf; Device<char>* d_win; Device<char>* d_members;
        Channel<int> free_bufs, full_bufs; int slot_id[2] = {0, 1}; size_t bytes[2] = {0, 0};
        hipEvent_t copied[2] = {nullptr, nullptr};        // the slot's window has reached its pinned buffer
        bool on_device = false;
        std::thread writer; std::exception_ptr error;
        explicit Out(SurvivorBuffers::PerFile& p) : src_off(p.src_off), dst_off(p.dst_off), len(p.len), total(p.total), buf(p.buf), d_win(p.d_win), d_members(p.d_members), on_device(p.on_device) {}
    };
    Out o[2] = {Out(planned->f[0]), Out(planned->f[1])};
    static int kStop = -1;
    // A window leaves the device on a stream of its own while the kernels of the next one run: the writer thread waits
    // for the copy, not this loop.  (A slot's device buffers are free again when its pinned buffer is: the writer gives
    // the slot back after it has written it.)
    hipStream_t down = nullptr;
    hipEvent_t made = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&made, hipEventDisableTiming));
    for (int s = 0; s < S; ++s) for (int k = 0; k < 2; ++k) HIP_OK(hipEventCreateWithFlags(&o[s].copied[k], hipEventDisableTiming));
    struct DownGuard { hipStream_t& d; hipEvent_t& m; Out* o; ~DownGuard() {
        if (d) { (void)hipStreamSynchronize(d); (void)hipStreamDestroy(d); }
        if (m) (void)hipEventDestroy(m);
        for (int s = 0; s < 2; ++s) for (int k = 0; k < 2; ++k) if (o[s].copied[k]) (void)hipEventDestroy(o[s].copied[k]);
    } } down_guard{down, made, o};
    for (int s = 0; s < S; ++s) {
        o[s].free_bufs.push(&o[s].slot_id[0]); o[s].free_bufs.push(&o[s].slot_id[1]);
        o[s].writer = std::thread([&, s] {
            for (;;) {
                int* id = o[s].full_bufs.pop();
                if (*id < 0) break;
                try {
                    { StageClock::Scope t("  survivors: writer waits for the window's copy"); HIP_OK(hipEventSynchronize(o[s].copied[*id])); }
                    StageClock::Scope t("  survivors: writer writes");
                    if (!o[s].error) {
                        if (o[s].on_device) sinks[s]->write_members(o[s].buf[*id].p, o[s].bytes[*id], write_threads());
                        else if (sinks[s]->is_gz()) sinks[s]->write_borrowed(o[s].buf[*id].p, o[s].bytes[*id]);
                        else {                                   // a plain file: the window in slices, copied in by several threads
                            constexpr size_t kSlices = 128;
                            OutputFile::Piece pieces[kSlices];
                            const size_t n = o[s].bytes[*id];
                            for (size_t k = 0; k < kSlices; ++k) { const size_t a = n / kSlices * k, b = k + 1 == kSlices ? n : n / kSlices * (k + 1); pieces[k] = {o[s].buf[*id].p + a, b - a}; }
                            sinks[s]->write_pieces(pieces, kSlices, write_threads());
                        }
                    }
                }
                catch (...) { o[s].error = std::current_exception(); }
                o[s].free_bufs.push(id);
            }
        });
    }
    auto peek_u64 = [&](const uint64_t* d, uint64_t k) {
        uint64_t v = 0;
        HIP_OK(hipMemcpyAsync(&v, d + k, sizeof v, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return v;
    };
    std::exception_ptr failure;
    try {
        uint64_t at[2] = {0, 0};
        while (at[0] < upto || (S == 2 && at[1] < upto)) {
            for (int s = 0; s < S; ++s) {
                if (at[s] >= upto) continue;
                // as many pairs as fill a window: from the average record size, halved until the bytes fit
                const uint64_t avg = std::max<uint64_t>(1, o[s].total / std::max<uint64_t>(1, upto - dups));
                uint64_t take = std::min<uint64_t>(upto - at[s], std::max<uint64_t>(1, window / avg));
                const uint64_t lo = peek_u64(o[s].dst_off.p, at[s]);
                uint64_t hi;
                for (;;) {
                    hi = at[s] + take == upto ? o[s].total : peek_u64(o[s].dst_off.p, at[s] + take);
                    if (hi - lo <= roomy || take == 1) break;
                    take = std::max<uint64_t>(1, take / 2);
                }
                const uint64_t bytes = hi - lo;
                if (bytes) {
                    int* id = nullptr;
                    { StageClock::Scope t("  survivors: the device waits for a free buffer"); id = o[s].free_bufs.pop(); }
                    StageClock::Scope t("  survivors: windows made on the device");
                    // every buffer is sized once, for the largest window the loop above lets through: a window a little
                    // larger than all before it must not cost a new pinned allocation (tens of milliseconds each)
                    const uint64_t room = std::max(bytes, std::min<uint64_t>(roomy, o[s].total));      // (a small output: what it needs)
                    Device<char>& d_win = o[s].d_win[*id];
                    d_win.reserve(room + 64);
                    o[s].buf[*id].reserve((o[s].on_device ? std::max<uint64_t>(room / 2, 1u << 20) : room) + 64);
                    // dst_off is absolute in the output: the window's buffer starts `lo` bytes in
                    engine_ok(fqd_copy_spans(e, reinterpret_cast<const uint8_t*>(file[s]->text.p), o[s].src_off.p + at[s], o[s].len.p + at[s], take,
                                             reinterpret_cast<uint8_t*>(d_win.p) - lo, o[s].dst_off.p + at[s]));
                    uint64_t out_bytes = bytes;
                    const char* from = d_win.p;
                    if (o[s].on_device) {
                        Device<char>& d_members = o[s].d_members[*id];
                        const uint64_t cap = fqd_bgzf_bound(room);
                        d_members.reserve(cap);
                        engine_ok(fqd_bgzf_deflate(e, reinterpret_cast<const uint8_t*>(d_win.p), bytes, lines_per_record,
                                                   reinterpret_cast<uint8_t*>(d_members.p), cap, &out_bytes));
                        from = d_members.p;
                        o[s].buf[*id].reserve(out_bytes + 64);           // (text that does not shrink to half)
                    }
                    HIP_OK(hipEventRecord(made, stream));
                    HIP_OK(hipStreamWaitEvent(down, made, 0));
                    HIP_OK(hipMemcpyAsync(o[s].buf[*id].p, from, out_bytes, hipMemcpyDeviceToHost, down));
                    HIP_OK(hipEventRecord(o[s].copied[*id], down));
                    o[s].bytes[*id] = out_bytes;
                    o[s].full_bufs.push(id);
                }
                at[s] += take;
            }
        }
    } catch (...) { failure = std::current_exception(); }
    for (int s = 0; s < S; ++s) { o[s].full_bufs.push(&kStop); o[s].writer.join(); }
    if (failure) std::rethrow_exception(failure);
    for (int s = 0; s < S; ++s) { if (o[s].error) std::rethrow_exception(o[s].error); if (close_sinks) sinks[s]->close(); }
}

} // namespace detail

} // namespace fqdhost
