#include "file_io.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <iostream>
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

OutputFile::OutputFile(const std::string& name) : gz_(has_gz_extension(name)), name_(name)
{
    if (gz_) {
        g_ = gzopen(name.c_str(), "wb");                       // file_utils.cpp:87-88 (64 KiB buffers)
        if (!g_) throw_cannot_open(name);
        gzbuffer(g_, 64 * 1024);
    } else {
        f_ = std::fopen(name.c_str(), "wb");                   // file_utils.cpp:90 (256 KiB buffer)
        if (!f_) throw_cannot_open(name);
        std::setvbuf(f_, nullptr, _IOFBF, 256 * 1024);
    }
}

OutputFile::~OutputFile() { close(); }

void OutputFile::write(const char* p, size_t n)
{
    while (n) {
        if (gz_) {
            const unsigned chunk = static_cast<unsigned>(std::min<size_t>(n, 1u << 30));
            if (gzwrite(g_, p, chunk) <= 0) throw std::runtime_error("write failed: " + name_);
            p += chunk; n -= chunk;
        } else {
            const size_t k = std::fwrite(p, 1, n, f_);
            if (k == 0) throw std::runtime_error("write failed: " + name_);
            p += k; n -= k;
        }
    }
}

void OutputFile::close()
{
    if (g_) { gzclose(g_); g_ = nullptr; }
    if (f_) { std::fclose(f_); f_ = nullptr; }
}

} // namespace fqdhost
