#include "file_io.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <algorithm>
#include <iostream>
#include <thread>
#include <unistd.h>

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
    }
}

InputFile::~InputFile()
{
    if (g_) gzclose(g_);
    if (fd_ >= 0) ::close(fd_);
}

size_t InputFile::read(char* dst, size_t n)
{
    size_t got = 0;
    while (got < n && !eof_) {
        if (gz_) {
            const unsigned want = static_cast<unsigned>(std::min<size_t>(n - got, 1u << 30));
            const int k = gzread(g_, dst + got, want);
            if (k < 0) throw std::runtime_error("gzip input is corrupt or truncated");
            if (k == 0) eof_ = true;
            got += static_cast<size_t>(k);
        } else {
            const ssize_t k = ::read(fd_, dst + got, n - got);
            if (k < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("read failed: ") + std::strerror(errno)); }
            if (k == 0) eof_ = true;
            got += static_cast<size_t>(k);
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
    f_ = std::fopen(name.c_str(), "wb");
    if (!f_) throw_cannot_open(name);
    std::setvbuf(f_, nullptr, _IOFBF, 256 * 1024);
    if (gz_) {
        block_.reserve(kGzMember + 65536);
        const unsigned hw = std::thread::hardware_concurrency();
        max_in_flight_ = hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1);
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

void OutputFile::write(const char* p, size_t n)
{
    if (!gz_) {
        while (n) {
            const size_t k = std::fwrite(p, 1, n, f_);
            if (k == 0) throw std::runtime_error("write failed: " + name_);
            p += k; n -= k;
        }
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

void OutputFile::close()
{
    if (!f_) return;
    if (gz_) {
        submit_block();
        drain(0);
        if (std::ftell(f_) == 0) {                          // nothing was written: still a valid (empty) gzip file
            const std::string member = deflate_member(std::string());
            std::fwrite(member.data(), 1, member.size(), f_);
        }
    }
    std::fclose(f_);
    f_ = nullptr;
}

} // namespace fqdhost
