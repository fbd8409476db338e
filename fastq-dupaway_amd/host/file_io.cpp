#include "file_io.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <algorithm>
#include <iostream>
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

InputFile::InputFile(const std::string& name) : gz_(has_gz_extension(name))
{
    if (gz_) {
        g_ = gzopen(name.c_str(), "rb");
        if (!g_) throw_cannot_open(name);
        gzbuffer(g_, 4u << 20);
    } else {
        fd_ = ::open(name.c_str(), O_RDONLY);
        if (fd_ < 0) throw_cannot_open(name);
#ifdef POSIX_FADV_SEQUENTIAL
        (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
        struct stat st;
        if (::fstat(fd_, &st) == 0 && S_ISREG(st.st_mode)) { regular_ = true; size_ = static_cast<uint64_t>(st.st_size); }
    }
}

InputFile::~InputFile()
{
    if (g_) gzclose(g_);
    if (fd_ >= 0) ::close(fd_);
}

size_t InputFile::read(char* dst, size_t n, unsigned threads)
{
    constexpr size_t kMinPart = 16u << 20;
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
            if (k == 0) eof_ = true;
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

constexpr size_t kGzMember = 1u << 20;                   // uncompressed bytes per gzip member

// One complete gzip member (header + deflate stream + CRC/length trailer) for `raw`.
std::string deflate_member(std::string raw)
{
    z_stream zs{};
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        throw std::runtime_error("zlib: deflateInit2 failed");
    std::string out(deflateBound(&zs, static_cast<uLong>(raw.size())) + 64, '\0');
    zs.next_in = reinterpret_cast<Bytef*>(raw.data());
    zs.avail_in = static_cast<uInt>(raw.size());
    zs.next_out = reinterpret_cast<Bytef*>(out.data());
    zs.avail_out = static_cast<uInt>(out.size());
    const int rc = deflate(&zs, Z_FINISH);
    const size_t produced = out.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("zlib: deflate failed");
    out.resize(produced);
    return out;
}

} // namespace

OutputFile::OutputFile(const std::string& name) : gz_(has_gz_extension(name)), name_(name)
{
    if (gz_) {
        f_ = std::fopen(name.c_str(), "wb");
        if (!f_) throw_cannot_open(name);
        std::setvbuf(f_, nullptr, _IOFBF, 256 * 1024);
        block_.reserve(kGzMember + 65536);
        const unsigned hw = std::thread::hardware_concurrency();
        max_in_flight_ = hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1);
    } else {
        fd_ = ::open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd_ < 0) throw_cannot_open(name);
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

void OutputFile::write_pieces(const Piece* pieces, size_t count)
{
    if (gz_) { for (size_t k = 0; k < count; ++k) write(pieces[k].p, pieces[k].n); return; }
    flush_plain();
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
        ::close(fd_);
        fd_ = -1;
        return;
    }
    if (!f_) return;
    submit_block();
    drain(0);
    if (std::ftell(f_) == 0) {                              // nothing was written: still a valid (empty) gzip file
        const std::string member = deflate_member(std::string());
        std::fwrite(member.data(), 1, member.size(), f_);
    }
    std::fclose(f_);
    f_ = nullptr;
}

} // namespace fqdhost
