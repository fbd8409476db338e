#include "file_io.hpp"
#include "pgzip.hpp"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <algorithm>
#include <iostream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace fqdhost {

bool has_gz_extension(const std::string& name)
{
    // std::filesystem::path(name).extension() == ".gz": text from the last '.' of the last
    // path component; a leading-dot-only name ("." / ".gz" as a hidden file) has no extension.
    const size_t slash = name.find_last_of('/');
    const std::string leaf = slash == std::string::npos ? name : name.substr(slash + 1);
    const size_t dot = leaf.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return false;
    return leaf.compare(dot, std::string::npos, ".gz") == 0;
}

void throw_cannot_open(const std::string& name)
{
    throw DiagnosedError("Cannot open file " + name + "\n", "File does not exist or cannot be opened!");
}

// gzip header with FLG = FEXTRA and a 'B','C' subfield holding (total size - 1).
size_t bgzf_member_size(const unsigned char* p, size_t avail, size_t* data_off)
{
    if (avail < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || p[3] != 4) return 0;
    const size_t xlen = p[10] | (size_t(p[11]) << 8);
    if (avail < 12 + xlen) return 0;
    for (size_t at = 12; at + 4 <= 12 + xlen;) {
        const size_t slen = p[at + 2] | (size_t(p[at + 3]) << 8);
        if (p[at] == 'B' && p[at + 1] == 'C' && slen == 2 && at + 6 <= 12 + xlen) {
            *data_off = 12 + xlen;
            const size_t total = (p[at + 4] | (size_t(p[at + 5]) << 8)) + 1;
            return total >= 12 + xlen + 8 ? total : 0;
        }
        at += 4 + slen;
    }
    return 0;
}

namespace {

// libdeflate (the whole-buffer DEFLATE library htslib also uses for BGZF) when the system has it: two to
// three times zlib's speed on 64 KB members, same format.  Looked up at run time by its soname — the image
// ships the library without its header, so the handful of entry points used are declared here as its
// documented C API has them.  Absent (or FQD_CODEC=zlib): zlib does the same work.
struct FastCodec {
    void* (*alloc_compressor)(int level) = nullptr;
    size_t (*compress)(void*, const void* in, size_t in_n, void* out, size_t out_avail) = nullptr;    // 0: did not fit
    void (*free_compressor)(void*) = nullptr;
    void* (*alloc_decompressor)() = nullptr;
    int (*decompress)(void*, const void* in, size_t in_n, void* out, size_t out_avail, size_t* actual) = nullptr;   // 0: ok
    void (*free_decompressor)(void*) = nullptr;
    uint32_t (*crc32)(uint32_t crc, const void* buf, size_t n) = nullptr;

    static const FastCodec* get()
    {
        static const FastCodec* codec = []() -> const FastCodec* {
            const char* pick = std::getenv("FQD_CODEC");
            if (pick && std::strcmp(pick, "zlib") == 0) return nullptr;
            void* h = ::dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) return nullptr;
            static FastCodec c;
            auto sym = [&](const char* name) { return ::dlsym(h, name); };
            c.alloc_compressor = reinterpret_cast<decltype(c.alloc_compressor)>(sym("libdeflate_alloc_compressor"));
            c.compress = reinterpret_cast<decltype(c.compress)>(sym("libdeflate_deflate_compress"));
            c.free_compressor = reinterpret_cast<decltype(c.free_compressor)>(sym("libdeflate_free_compressor"));
            c.alloc_decompressor = reinterpret_cast<decltype(c.alloc_decompressor)>(sym("libdeflate_alloc_decompressor"));
            c.decompress = reinterpret_cast<decltype(c.decompress)>(sym("libdeflate_deflate_decompress"));
            c.free_decompressor = reinterpret_cast<decltype(c.free_decompressor)>(sym("libdeflate_free_decompressor"));
            c.crc32 = reinterpret_cast<decltype(c.crc32)>(sym("libdeflate_crc32"));
            if (!c.alloc_compressor || !c.compress || !c.free_compressor || !c.alloc_decompressor || !c.decompress ||
                !c.free_decompressor || !c.crc32) return nullptr;
            return &c;
        }();
        return codec;
    }
};

} // namespace

const char* gz_codec_name() { return FastCodec::get() ? "libdeflate" : "zlib"; }

InputFile::InputFile(const std::string& name, bool as_bytes) : gz_(!as_bytes && has_gz_extension(name))
{
    fd_ = ::open(name.c_str(), O_RDONLY);
    if (fd_ < 0) throw_cannot_open(name);
#ifdef POSIX_FADV_SEQUENTIAL
    (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    struct stat st;
    if (::fstat(fd_, &st) == 0 && S_ISREG(st.st_mode)) { regular_ = true; size_ = static_cast<uint64_t>(st.st_size); }
    if (gz_) {
        unsigned char head[64];
        size_t off = 0;
        const ssize_t k = regular_ ? ::pread(fd_, head, sizeof head, 0) : 0;
        if (k >= 18 && bgzf_member_size(head, static_cast<size_t>(k), &off) != 0) {
            bgzf_ = true;
        } else {
            // an ordinary gzip file: by several threads (pgzip.hpp) if the first read asks for that and the file is a
            // regular one of some size; through zlib's gzread otherwise.  Decided at the first read.
            undecided_ = true;
        }
    }
}

InputFile::~InputFile()
{
    delete static_cast<pgz::Reader*>(pgzip_);
    if (g_) gzclose(g_);
    if (fd_ >= 0) ::close(fd_);
}

// Tops the compressed buffer up; false when nothing more can come.
bool InputFile::fill_compressed()
{
    if (comp_eof_) return false;
    if (comp_pos_ > 0) { comp_.erase(comp_.begin(), comp_.begin() + static_cast<ptrdiff_t>(comp_pos_)); comp_pos_ = 0; }
    constexpr size_t kChunk = 16u << 20;
    const size_t old = comp_.size();
    comp_.resize(old + kChunk);
    size_t got = 0;
    while (got < kChunk) {
        const ssize_t k = ::pread(fd_, comp_.data() + old + got, kChunk - got, static_cast<off_t>(offset_));
        if (k < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("read failed: ") + std::strerror(errno)); }
        if (k == 0) { comp_eof_ = true; break; }
        got += static_cast<size_t>(k); offset_ += static_cast<uint64_t>(k);
    }
    comp_.resize(old + got);
    return got > 0;
}

size_t InputFile::read_bgzf(char* dst, size_t n, unsigned threads)
{
    struct Member { size_t at, data_off, total, isize; };
    // 1 = a whole member lies at `at`, 0 = more compressed bytes are needed, -1 = not a BGZF member
    auto member_at = [&](size_t at, Member& m) -> int {
        const size_t avail = comp_.size() - at;
        const unsigned char* p = comp_.data() + at;
        if (avail < 18) return 0;
        if (p[0] != 31 || p[1] != 139 || p[2] != 8 || p[3] != 4) return -1;
        if (avail < 12 + (p[10] | (size_t(p[11]) << 8))) return 0;
        size_t off = 0;
        const size_t total = bgzf_member_size(p, avail, &off);
        if (total == 0) return -1;
        if (avail < total) return 0;
        const unsigned char* tail = p + total - 4;
        m = Member{at, off, total, tail[0] | (size_t(tail[1]) << 8) | (size_t(tail[2]) << 16) | (size_t(tail[3]) << 24)};
        if (m.isize > 65536) return -1;                       // a BGZF member never holds more: a damaged trailer, not a size to allocate
        return 1;
    };
    size_t got = 0;
    while (got < n && !eof_) {
        if (spill_pos_ < spill_.size()) {                        // what the previous call could not take
            const size_t k = std::min(n - got, spill_.size() - spill_pos_);
            std::memcpy(dst + got, spill_.data() + spill_pos_, k);
            spill_pos_ += k; got += k;
            continue;
        }
        // cut a batch of whole members out of the compressed buffer
        std::vector<Member> batch;
        size_t out_bytes = 0, at = comp_pos_;
        for (;;) {
            Member m{};
            const int r = member_at(at, m);
            if (r < 0) {
                // A gzip member that is no BGZF member, after some that were (files put together with `cat`): the
                // reference's gzip_decompressor reads members of any kind one after the other (file_utils.hpp:58-69),
                // so from here on the stream goes through zlib, which does the same.  Anything else is damage.
                const unsigned char* q = comp_.data() + at;
                size_t ignored = 0;
                const bool plain_member = comp_.size() - at >= 18 && q[0] == 31 && q[1] == 139 && q[2] == 8 &&
                                          bgzf_member_size(q, comp_.size() - at, &ignored) == 0;
                if (!plain_member) throw std::runtime_error("gzip input is corrupt or truncated");
                if (!batch.empty()) break;                       // first what came before it
                const uint64_t file_at = offset_ - (comp_.size() - at);
                if (::lseek(fd_, static_cast<off_t>(file_at), SEEK_SET) < 0) throw std::runtime_error("gzip input is corrupt or truncated");
                g_ = gzdopen(fd_, "rb");
                if (!g_) throw std::runtime_error("gzip input is corrupt or truncated");
                fd_ = -1;                                          // (the stream owns the descriptor now)
                gzbuffer(g_, 4u << 20);
                bgzf_ = false; comp_.clear(); comp_pos_ = 0;
                return got + read(dst + got, n - got, threads);
            }
            if (r == 1) {
                batch.push_back(m); out_bytes += m.isize; at += m.total;
                if (out_bytes >= n - got || batch.size() >= 4096) break;
                continue;
            }
            if (!batch.empty()) break;                           // inflate what is here before fetching more
            const size_t pending = comp_.size() - comp_pos_;
            if (!fill_compressed()) {
                if (pending != 0) throw std::runtime_error("gzip input is corrupt or truncated");
                eof_ = true;
                break;
            }
            at = comp_pos_;
        }
        if (batch.empty()) continue;
        // members that fit go straight to dst; one that straddles the end goes to the spill buffer
        std::vector<char*> target(batch.size(), nullptr);
        size_t pos = got, used = 0;
        spill_.clear(); spill_pos_ = 0;
        for (; used < batch.size(); ++used) {
            if (pos + batch[used].isize <= n) { target[used] = dst + pos; pos += batch[used].isize; }
            else { spill_.resize(batch[used].isize); target[used] = spill_.data(); ++used; break; }
        }
        batch.resize(used);
        const unsigned parts = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(threads, batch.size() / 8)));
        std::vector<int> bad(parts, 0);
        const FastCodec* fast = FastCodec::get();
        auto work = [&](unsigned p) {
            if (fast) {
                void* d = fast->alloc_decompressor();
                if (!d) { bad[p] = 1; return; }
                for (size_t k = p; k < batch.size(); k += parts) {
                    const Member& m = batch[k];
                    if (m.isize == 0) continue;
                    const unsigned char* src = comp_.data() + m.at;
                    const unsigned char* t = src + m.total - 8;
                    const uint32_t want_crc = t[0] | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
                    size_t actual = 0;
                    if (fast->decompress(d, src + m.data_off, m.total - m.data_off - 8, target[k], m.isize, &actual) != 0 ||
                        actual != m.isize || fast->crc32(0, target[k], m.isize) != want_crc) { bad[p] = 1; break; }
                }
                fast->free_decompressor(d);
                return;
            }
            z_stream zs{};
            if (inflateInit2(&zs, -15) != Z_OK) { bad[p] = 1; return; }
            for (size_t k = p; k < batch.size(); k += parts) {
                const Member& m = batch[k];
                if (m.isize == 0) continue;                       // the end-of-file marker, or an empty flush
                const unsigned char* src = comp_.data() + m.at;
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef*>(src + m.data_off);
                zs.avail_in = static_cast<uInt>(m.total - m.data_off - 8);
                zs.next_out = reinterpret_cast<Bytef*>(target[k]);
                zs.avail_out = static_cast<uInt>(m.isize);
                const int rc = inflate(&zs, Z_FINISH);
                const unsigned char* t = src + m.total - 8;
                const uLong want_crc = t[0] | (uLong(t[1]) << 8) | (uLong(t[2]) << 16) | (uLong(t[3]) << 24);
                if (rc != Z_STREAM_END || zs.avail_out != 0 ||
                    crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(target[k]), static_cast<uInt>(m.isize)) != want_crc) { bad[p] = 1; break; }
            }
            inflateEnd(&zs);
        };
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < parts; ++p) pool.emplace_back(work, p);
        work(0);
        for (std::thread& t : pool) t.join();
        for (unsigned p = 0; p < parts; ++p) if (bad[p]) throw std::runtime_error("gzip input is corrupt or truncated");
        comp_pos_ = batch.back().at + batch.back().total;
        got = pos;
    }
    return got;
}

size_t InputFile::read(char* dst, size_t n, unsigned threads)
{
    if (undecided_) {
        undecided_ = false;
        static const uint64_t least = [] { const char* v = std::getenv("FQD_PGZIP_MIN_MB"); return (v ? uint64_t(std::atoll(v)) : 8u) << 20; }();
        static const bool allowed = [] { const char* v = std::getenv("FQD_PGZIP"); return !v || std::atoi(v) != 0; }();
        if (allowed && regular_ && threads >= 2 && size_ >= least) {
            try { const FastCodec* fast = FastCodec::get(); pgzip_ = new pgz::Reader(fd_, size_, threads, fast ? fast->crc32 : nullptr); }
            catch (const std::invalid_argument&) { pgzip_ = nullptr; }            // (a header it does not know: zlib reads the file)
        }
        if (!pgzip_) {
            g_ = gzdopen(fd_, "rb");                                              // (the descriptor has only been pread so far: it stands at 0)
            if (!g_) throw std::runtime_error("gzip input is corrupt or truncated");
            fd_ = -1;
            gzbuffer(g_, 4u << 20);
        }
    }
    if (pgzip_) {
        if (eof_) return 0;
        const size_t got = static_cast<pgz::Reader*>(pgzip_)->read(dst, n);
        if (got < n) eof_ = true;
        return got;
    }
    if (bgzf_) return read_bgzf(dst, n, threads);
    static const size_t kMinPart = [] { const char* v = std::getenv("FQD_READ_PART_MB"); const long mb = v ? std::atol(v) : 0; return size_t(mb > 0 ? mb : 16) << 20; }();   // bytes per reading thread
    if (!gz_ && regular_ && threads > 1 && n >= 2 * kMinPart && !eof_) {
        const size_t want = static_cast<size_t>(std::min<uint64_t>(n, size_ > offset_ ? size_ - offset_ : 0));
        const unsigned parts = static_cast<unsigned>(std::min<size_t>(threads, std::max<size_t>(1, want / kMinPart)));
        std::vector<size_t> done(parts, 0);
        std::vector<int> err(parts, 0);
        auto part = [&](unsigned p) {
            const size_t lo = want / parts * p, hi = p + 1 == parts ? want : want / parts * (p + 1);
            size_t at = lo;
            while (at < hi) {
                const ssize_t k = ::pread(fd_, dst + at, hi - at, static_cast<off_t>(offset_ + at));
                if (k < 0) { if (errno == EINTR) continue; err[p] = errno; break; }
                if (k == 0) break;                               // the file shrank under us
                at += static_cast<size_t>(k);
            }
            done[p] = at - lo;
        };
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < parts; ++p) pool.emplace_back(part, p);
        part(0);
        for (std::thread& t : pool) t.join();
        size_t got = 0;
        for (unsigned p = 0; p < parts; ++p) {
            if (err[p]) throw std::runtime_error(std::string("read failed: ") + std::strerror(err[p]));
            const size_t lo = want / parts * p, hi = p + 1 == parts ? want : want / parts * (p + 1);
            got += done[p];
            if (done[p] != hi - lo) { eof_ = true; break; }       // short part: nothing after it counts
        }
        offset_ += got;
        if (got < n) {                                            // end of file (or a short part): confirm with a plain read
            const ssize_t k = ::pread(fd_, dst + got, n - got, static_cast<off_t>(offset_));
            if (k > 0) { got += static_cast<size_t>(k); offset_ += static_cast<uint64_t>(k); }
            if (got < n) eof_ = true;
        }
        return got;
    }
    size_t got = 0;
    while (got < n && !eof_) {
        if (gz_) {
            const unsigned want = static_cast<unsigned>(std::min<size_t>(n - got, 1u << 30));
            const int k = gzread(g_, dst + got, want);
            if (k < 0) throw std::runtime_error("gzip input is corrupt or truncated");
            if (k == 0) {
                // zlib hands out what a stream that ends too early held and says so only now (Z_BUF_ERROR at end of file): the
                // reference's decompressor throws on such a file (file_utils.cpp:59-66), a clean end of input this is not
                int en = Z_OK;
                (void)gzerror(g_, &en);
                if (en == Z_BUF_ERROR || en == Z_DATA_ERROR) throw std::runtime_error("gzip input is corrupt or truncated");
                eof_ = true;
            }
            got += static_cast<size_t>(k);
        } else {
            const ssize_t k = regular_ ? ::pread(fd_, dst + got, n - got, static_cast<off_t>(offset_)) : ::read(fd_, dst + got, n - got);
            if (k < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("read failed: ") + std::strerror(errno)); }
            if (k == 0) eof_ = true;
            got += static_cast<size_t>(k); offset_ += static_cast<uint64_t>(k);
        }
    }
    return got;
}

namespace {

constexpr size_t kBgzfInput = 0xff00;                     // input bytes per BGZF member (as bgzip)
constexpr size_t kGzMember = 16 * kBgzfInput;             // input bytes per deflate job: 16 members

// The BGZF members (header with the 'BC' size field + raw deflate stream + CRC/length trailer)
// for `raw`, one per kBgzfInput bytes.
std::string deflate_range(const char* raw, size_t raw_size);
std::string deflate_member(std::string raw) { return deflate_range(raw.data(), raw.size()); }

// The same over bytes that stay where they are (they must outlive the job).
std::string deflate_range(const char* raw, size_t raw_size)
{
    std::string out;
    out.reserve(raw_size / 3 + 1024);
    z_stream zs{};
    // zlib's default level, as the reference's Boost gzip filter uses; FQD_GZ_LEVEL=1..9 trades size for speed
    static const int level = [] { const char* v = std::getenv("FQD_GZ_LEVEL"); const int l = v ? std::atoi(v) : 0; return l >= 1 && l <= 9 ? l : Z_DEFAULT_COMPRESSION; }();
    const FastCodec* fast = FastCodec::get();
    void* fc = fast ? fast->alloc_compressor(level == Z_DEFAULT_COMPRESSION ? 6 : level) : nullptr;
    bool zs_ready = false;
    unsigned char body[65536];
    for (size_t at = 0; at < raw_size; at += kBgzfInput) {
        const size_t len = std::min(kBgzfInput, raw_size - at);
        size_t clen = fc ? fast->compress(fc, raw + at, len, body, sizeof body - 26) : 0;
        if (clen == 0) {                                         // no libdeflate, or its output did not fit a member
            if (!zs_ready) {
                if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                    if (fc) fast->free_compressor(fc);
                    throw std::runtime_error("zlib: deflateInit2 failed");
                }
                zs_ready = true;
            }
            deflateReset(&zs);
            zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(raw + at));
            zs.avail_in = static_cast<uInt>(len);
            zs.next_out = body;
            zs.avail_out = static_cast<uInt>(sizeof body - 26);
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); if (fc) fast->free_compressor(fc); throw std::runtime_error("zlib: deflate failed"); }
            clen = sizeof body - 26 - zs.avail_out;
        }
        const size_t total = 18 + clen + 8;
        const uLong crc = fast ? fast->crc32(0, raw + at, len) : crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(raw + at), static_cast<uInt>(len));
        const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0,
                                        static_cast<unsigned char>((total - 1) & 0xFF), static_cast<unsigned char>((total - 1) >> 8)};
        const unsigned char tail[8] = {static_cast<unsigned char>(crc), static_cast<unsigned char>(crc >> 8),
                                       static_cast<unsigned char>(crc >> 16), static_cast<unsigned char>(crc >> 24),
                                       static_cast<unsigned char>(len), static_cast<unsigned char>(len >> 8), 0, 0};
        out.append(reinterpret_cast<const char*>(head), sizeof head);
        out.append(reinterpret_cast<const char*>(body), clen);
        out.append(reinterpret_cast<const char*>(tail), sizeof tail);
    }
    if (zs_ready) deflateEnd(&zs);
    if (fc) fast->free_compressor(fc);
    return out;
}

// The empty member every BGZF file ends with.
const unsigned char kBgzfEof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

} // namespace

OutputFile::OutputFile(const std::string& name) : gz_(has_gz_extension(name)), name_(name)
{
    if (gz_) {
        f_ = std::fopen(name.c_str(), "wb");
        if (!f_) throw_cannot_open(name);
        std::setvbuf(f_, nullptr, _IOFBF, 256 * 1024);
        block_.reserve(kGzMember + 65536);
        const unsigned hw = std::thread::hardware_concurrency();
        max_in_flight_ = hw >= 32 ? 16 : (hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1));    // deflate jobs in flight (16 BGZF members each)
        if (const char* v = std::getenv("FQD_GZ_JOBS")) { const int x = std::atoi(v); if (x > 0) max_in_flight_ = static_cast<size_t>(x); }
    } else {
        fd_ = ::open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd_ < 0) throw_cannot_open(name);
        struct stat st;
        regular_ = ::fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
        plain_buf_.reserve(256 * 1024);
    }
}

OutputFile::~OutputFile()
{
    try { close(); } catch (...) {}
}

void OutputFile::drain(size_t keep_in_flight)
{
    while (in_flight_.size() > keep_in_flight) {
        const std::string member = in_flight_.front().get();
        in_flight_.pop_front();
        if (!member.empty() && std::fwrite(member.data(), 1, member.size(), f_) != member.size())
            throw std::runtime_error("write failed: " + name_);
    }
}

void OutputFile::submit_block()
{
    if (block_.empty()) return;
    std::string raw;
    raw.swap(block_);
    block_.reserve(kGzMember + 65536);
    in_flight_.push_back(std::async(std::launch::async, deflate_member, std::move(raw)));
    drain(max_in_flight_);
}

// A large buffer the caller lends until the call returns: gzip members are deflated straight out of it on
// the worker threads (write() would first copy every byte into the member being filled).
void OutputFile::write_borrowed(const char* p, size_t n)
{
    if (!gz_ || n < 4 * kGzMember) { write(p, n); return; }
    submit_block();                                           // a member may be short: what was pending goes first
    for (size_t at = 0; at < n; at += kGzMember) {
        in_flight_.push_back(std::async(std::launch::async, deflate_range, p + at, std::min(kGzMember, n - at)));
        drain(max_in_flight_);
    }
    drain(0);                                                 // nothing refers to the buffer any more
}

void OutputFile::write_members(const char* p, size_t n, unsigned threads)
{
    if (!gz_) throw std::logic_error("write_members: not a .gz output");
    submit_block();
    drain(0);
    if (n == 0) return;
    // A large batch into a regular file: its place is reserved and mapped, and several threads copy their share
    // in — the page cache is filled in parallel, which one write stream cannot do (see write_pieces).
    if (threads > 1 && n >= (8u << 20) && std::fflush(f_) == 0) {
        const int fd = ::fileno(f_);
        struct stat st;
        const off_t at = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) ? ::lseek(fd, 0, SEEK_CUR) : off_t(-1);
        if (at >= 0 && ::fallocate(fd, 0, at, static_cast<off_t>(n)) == 0) {
            static const uint64_t page = static_cast<uint64_t>(::sysconf(_SC_PAGESIZE));
            const uint64_t lo = static_cast<uint64_t>(at) / page * page, lead = static_cast<uint64_t>(at) - lo;
            void* map = ::mmap(nullptr, lead + n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, static_cast<off_t>(lo));
            if (map != MAP_FAILED) {
                char* dst = static_cast<char*>(map) + lead;
                const unsigned T = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>(threads, n >> 21)));
                auto part = [&](unsigned k) { const size_t a = n / T * k, b = k + 1 == T ? n : n / T * (k + 1); std::memcpy(dst + a, p + a, b - a); };
                std::vector<std::thread> pool;
                for (unsigned k = 1; k < T; ++k) pool.emplace_back(part, k);
                part(0);
                for (std::thread& th : pool) th.join();
                ::munmap(map, lead + n);
                // the stream itself is told where the file now ends (moving the descriptor behind stdio's back
                // only works while glibc does not re-seek a write-only stream)
                if (::fseeko(f_, at + static_cast<off_t>(n), SEEK_SET) != 0) throw std::runtime_error("write failed: " + name_);
                return;
            }
        }
    }
    if (std::fwrite(p, 1, n, f_) != n) throw std::runtime_error("write failed: " + name_);
}

void OutputFile::put_plain(const char* p, size_t n)
{
    while (n) {
        const ssize_t k = ::write(fd_, p, n);
        if (k < 0) { if (errno == EINTR) continue; throw std::runtime_error("write failed: " + name_); }
        p += k; n -= static_cast<size_t>(k);
    }
}

void OutputFile::flush_plain()
{
    if (!plain_buf_.empty()) { put_plain(plain_buf_.data(), plain_buf_.size()); plain_buf_.clear(); }
}

void OutputFile::write(const char* p, size_t n)
{
    if (!gz_) {
        if (n >= 64 * 1024) { flush_plain(); put_plain(p, n); return; }
        if (plain_buf_.size() + n > 256 * 1024) flush_plain();
        plain_buf_.append(p, n);
        return;
    }
    while (n) {
        const size_t room = kGzMember > block_.size() ? kGzMember - block_.size() : 0;
        const size_t k = std::min(n, std::max<size_t>(room, 1));
        block_.append(p, k);
        p += k; n -= k;
        if (block_.size() >= kGzMember) submit_block();
    }
}

void OutputFile::write_pieces(const Piece* pieces, size_t count, unsigned threads)
{
    if (gz_) { for (size_t k = 0; k < count; ++k) write(pieces[k].p, pieces[k].n); return; }
    flush_plain();
    // A large batch into a regular file.  write()/pwrite() calls on ONE file are serialised by the
    // kernel (the inode lock), so several writer threads gain nothing that way — measured: slower.  The
    // batch's place in the file is known, though: reserve it (fallocate: a full disk is reported
    // here, as an error), map it, and let several threads copy their share of the pieces into the mapping;
    // page-cache pages are then filled in parallel (measured on the GPU box, 5.4 GB of survivors: 1.26 s
    // against 1.4-1.5 s for one writev stream; leaving the reservation out — ftruncate — was no faster).
    // Anything that does not support this falls through to the plain writev below.
    if (threads > 1 && regular_ && count >= 64) {
        uint64_t total = 0;
        for (size_t k = 0; k < count; ++k) total += pieces[k].n;
        const off_t at = total >= (8u << 20) ? ::lseek(fd_, 0, SEEK_CUR) : off_t(-1);
        // fallocate, not posix_fallocate: where the file system cannot reserve, glibc's emulation would write zeros first
        if (at >= 0 && ::fallocate(fd_, 0, at, static_cast<off_t>(total)) == 0) {
            static const uint64_t page = static_cast<uint64_t>(::sysconf(_SC_PAGESIZE));
            const uint64_t lo = static_cast<uint64_t>(at) / page * page, lead = static_cast<uint64_t>(at) - lo;
            void* map = ::mmap(nullptr, lead + total, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, static_cast<off_t>(lo));
            if (map != MAP_FAILED) {
                char* dst = static_cast<char*>(map) + lead;
                const unsigned T = static_cast<unsigned>(std::min<uint64_t>(threads, total >> 21));
                std::vector<size_t> first(T + 1, count); std::vector<uint64_t> off(T + 1, total);
                uint64_t run = 0; unsigned t = 0;
                for (size_t k = 0; k < count && t < T; ++k) {
                    if (run >= total / T * t) { first[t] = k; off[t] = run; ++t; }
                    run += pieces[k].n;
                }
                auto part = [&](unsigned p) {
                    char* q = dst + off[p];
                    for (size_t k = first[p]; k < first[p + 1]; ++k) { std::memcpy(q, pieces[k].p, pieces[k].n); q += pieces[k].n; }
                };
                std::vector<std::thread> pool;
                for (unsigned p = 1; p < T; ++p) pool.emplace_back(part, p);
                part(0);
                for (std::thread& th : pool) th.join();
                ::munmap(map, lead + total);
                if (::lseek(fd_, at + static_cast<off_t>(total), SEEK_SET) < 0) throw std::runtime_error("write failed: " + name_);
                return;
            }
        }
    }
    constexpr size_t kBatch = 1024;                          // IOV_MAX
    struct iovec iov[kBatch];
    size_t k = 0;
    while (k < count) {
        const size_t m = std::min(kBatch, count - k);
        for (size_t j = 0; j < m; ++j) { iov[j].iov_base = const_cast<char*>(pieces[k + j].p); iov[j].iov_len = pieces[k + j].n; }
        size_t first = 0;
        while (first < m) {                                  // writev may stop anywhere
            const ssize_t w = ::writev(fd_, iov + first, static_cast<int>(m - first));
            if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error("write failed: " + name_); }
            size_t left = static_cast<size_t>(w);
            while (first < m && left >= iov[first].iov_len) { left -= iov[first].iov_len; ++first; }
            if (first < m && left) { iov[first].iov_base = static_cast<char*>(iov[first].iov_base) + left; iov[first].iov_len -= left; }
        }
        k += m;
    }
}

void OutputFile::close()
{
    if (!gz_) {
        if (fd_ < 0) return;
        flush_plain();
        const int fd = fd_;
        fd_ = -1;
        if (::close(fd) != 0) throw std::runtime_error("write failed: " + name_);      // a full disk may only show here
        return;
    }
    if (!f_) return;
    submit_block();
    drain(0);
    FILE* f = f_;
    f_ = nullptr;
    const bool ok = std::fwrite(kBgzfEof, 1, sizeof kBgzfEof, f) == sizeof kBgzfEof && std::fflush(f) == 0;
    if (std::fclose(f) != 0 || !ok) throw std::runtime_error("write failed: " + name_);   // the last buffer and the EOF member go out here
}

} // namespace fqdhost
