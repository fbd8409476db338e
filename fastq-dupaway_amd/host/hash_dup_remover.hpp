// hash_dup_remover.hpp — host driver of the `--fast` path.
// Same interface as the reference's HashDupRemover<T> (src/hash_dup_remover.hpp:73-94):
//   HashDupRemover(memlimit, tempdir, verbose); filterSE(in, out);
//   filterPE(in1, in2, out1, out2, unordered)
// with the record type T (FastqView / FastaView [+WithId]) as a runtime Format.
// Records are parsed on the host (records.hpp), their raw blocks are DMA'd to HBM,
// the engine behind include/fqdupaway.h answers one keep flag per record (pair), and
// survivors are written verbatim in input order (tag order for --unordered).
#pragma once
#include <cstdint>
#include <string>
#include <sys/types.h>
#include <vector>

#include "records.hpp"

namespace fqdhost {

// Random 10-character directory in the CWD, removed on destruction
// (FileUtils::TemporaryDirectory, src/file_utils.hpp:96-108, src/file_utils.cpp:26-40,116-130).
// The GPU path keeps its intermediates in memory, so the directory is only created when
// somebody asks for its name.
class TemporaryDirectory {
public:
    TemporaryDirectory() = default;
    ~TemporaryDirectory();
    TemporaryDirectory(const TemporaryDirectory&) = delete;
    TemporaryDirectory& operator=(const TemporaryDirectory&) = delete;
    const char* name();
private:
    std::string name_;
};

struct Summary {                     // what -v prints (hash_dup_remover.hpp:146-147,253-254,342-346)
    uint64_t total = 0, duplicates = 0, unmatched = 0;
};

struct Tuning {
    int    device = 0;               // HIP device ordinal
    size_t block_bytes = 32u << 20;  // input block size per file (measured, 30 M reads plain: 16 MB 1.28 s, 32 MB 1.16 s, 64 MB 1.46 s, 128 MB 1.38 s)
    // --unordered: true = the reference's merge-join including its end-of-file rule
    // (hash_dup_remover.hpp:281,317-340), false = full inner join.
    bool   reference_tail_rule = true;
    // FQD_DEVICES=0,1,...: one engine per listed GPU, reads sharded by hash prefix with one all-to-all per
    // round (multi_gpu.hpp).  Empty: the single-engine path on `device`.  use_rccl = false (FQD_EXCHANGE=copy):
    // peer copies instead of RCCL.
    std::vector<int> devices;
    bool   use_rccl = true;
    // The process ends right after the run (the CLI): once the outputs of a resident run are closed, its tens of
    // gigabytes of HBM and pinned memory are left to process exit instead of being unmapped buffer by buffer.
    bool   leave_memory_to_exit = false;
};

class HashDupRemover {
public:
    HashDupRemover(Format format, ssize_t memlimit, TemporaryDirectory* tempdir, bool verbose, Tuning tuning = Tuning())
        : format_(format), memlimit_(memlimit), tempdir_(tempdir), verbose_(verbose), tuning_(tuning) {}
    void filterSE(const std::string& infile, const std::string& outfile);
    void filterPE(const std::string& infile1, const std::string& infile2,
                  const std::string& outfile1, const std::string& outfile2, bool unordered);
    const Summary& summary() const { return summary_; }
private:
    void run_ordered(int n_files, const std::string* in, const std::string* out);
    bool run_ordered_resident(int n_files, const std::string* in, const std::string* out);   // false: not taken, nothing touched
    void run_ordered_multi(int n_files, const std::string* in, const std::string* out);
    void run_unordered(const std::string* in, const std::string* out);
    void run_unordered_in_memory(const std::string* in, const std::string* out);
    void run_unordered_resident(const std::string* in, const std::string* out);
    void run_unordered_streaming(const std::string* in, const std::string* out);
    void run_unordered_multi(const std::string* in, const std::string* out);
    Format              format_;
    ssize_t             memlimit_;
    TemporaryDirectory* tempdir_;
    bool                verbose_;
    Tuning              tuning_;
    Summary             summary_;
};

} // namespace fqdhost
