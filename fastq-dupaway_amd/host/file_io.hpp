// file_io.hpp — plain / gzip input and output files for the host driver.
// Mirrors FileUtils::I_InputFile{TXT,GZ}, openInputFile and UniversalOutputFile of the
// reference (src/file_utils.hpp:25-79, src/file_utils.cpp:53-92): a ".gz" extension selects
// gzip, anything else is plain; same diagnostics when a file cannot be opened.
#pragma once
#include <cstddef>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <zlib.h>

namespace fqdhost {

bool has_gz_extension(const std::string& name);               // file_utils.cpp:42-48

// An error that comes with a line for stderr ahead of the exception text, the way the reference
// prints a diagnostic and then throws.  Whoever lets it reach main() prints `diag` first, so
// work done on helper threads never prints out of order.
struct DiagnosedError : std::runtime_error {
    std::string diag;
    DiagnosedError(std::string d, const std::string& what) : std::runtime_error(what), diag(std::move(d)) {}
};

// "Cannot open file <name>" + "File does not exist or cannot be opened!", as check_fstream_ok
// (file_utils.hpp:111-121).
[[noreturn]] void throw_cannot_open(const std::string& name);

class InputFile {
public:
    explicit InputFile(const std::string& name);
    ~InputFile();
    InputFile(const InputFile&) = delete;
    InputFile& operator=(const InputFile&) = delete;
    // Reads up to n bytes; fewer only at end of file (eof() turns true).
    size_t read(char* dst, size_t n);
    bool eof() const { return eof_; }
private:
    bool gz_; gzFile g_ = nullptr; int fd_ = -1; bool eof_ = false;
};

class OutputFile {
public:
    explicit OutputFile(const std::string& name);
    ~OutputFile();
    OutputFile(const OutputFile&) = delete;
    OutputFile& operator=(const OutputFile&) = delete;
    void write(const char* p, size_t n);
    void close();
private:
    bool gz_; gzFile g_ = nullptr; FILE* f_ = nullptr; std::string name_;
};

} // namespace fqdhost
