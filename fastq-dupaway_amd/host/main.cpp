// fastq-dupaway (MI355X-native build of the hash-based `--fast` mode).
// Same command line, messages and exit codes as the reference's main.cpp
// (src/main.cpp:40-262); the options are parsed the way Boost.program_options parses
// them there (long/short names, --name=value, -xVALUE, unambiguous long prefixes).
// Sequence-based modes are not part of this build and are refused with a clear message.
// Environment (extensions, none needed): FQD_DEVICE=<ordinal>, FQD_FULL_JOIN=1 (intended
// full inner join for --unordered instead of the reference's end-of-file rule),
// FQD_BLOCK_MB=<input block size>, FQD_DEVICES=<ordinal,ordinal,...> (one engine per listed GPU, reads
// sharded by hash prefix with an RCCL all-to-all; FQD_EXCHANGE=copy: peer copies instead).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "hash_dup_remover.hpp"
#include "multi_gpu.hpp"

namespace {

const char* kVersion = "fastq-dupaway V1.5.0";          // constants.hpp:10 (same CLI generation)

struct OptSpec { const char* longname; char shortname; bool takes_value; const char* help; };

const OptSpec kSpecs[] = {
    {"help", 'h', false, "Produce help message and exit"},
    {"verbose", 'v', false, "Report run summary after program execution."},
    {"input-1", 'i', true, "First input file (required)"},
    {"input-2", 'u', true, "Second input file (optional, enables paired-end mode)"},
    {"output-1", 'o', true, "First output file (required)"},
    {"output-2", 'p', true, "Second output file (optional, required for paired-end mode)"},
    {"mem-limit", 'm', true, "Memory limit in megabytes (default 2048 = 2Gb).\nSupported value range is [500 <-> 10240 (10 Gb)]\n"
                             "Actual memory usage may slightly exceed this value.\n"
                             "NB: The 'fast' deduplication mode does not support strict memory limitation."},
    {"format", 0, true, "input file format: fastq (default) or fasta."},
    {"compare-seq", 0, true, "Sequence comparison mode for deduplication step (sequence-based modes; not part of the MI355X build)."},
    {"distance", 0, true, "A threshold value for 'tail-hamming' distance calculation (sequence-based modes)."},
    {"write-clusters", 0, false, "Write ids of identified duplicate clusters to a file (sequence-based modes only)."},
    {"fast", 0, false, "Use hash-based approach instead of sequence-based.\nIn this mode the program will run significantly faster, "
                       "however no memory limit can be set and only complete duplicates will be filtered out."},
    {"unordered", 0, false, "This option is supported only by 'fast' mode for paired inputs.\nEnable this flag if reads in your paired "
                            "input files are not synchronized.\nIf this option is enabled, both input files will be joined by read IDs "
                            "before deduplication."},
};

void print_help()
{
    std::cerr << kVersion << "\n";
    std::cerr << "Supported options:\n";
    for (const OptSpec& o : kSpecs) {
        std::string left = "  ";
        if (o.shortname) { left += '-'; left += o.shortname; left += " [ --"; left += o.longname; left += " ]"; }
        else             { left += "--"; left += o.longname; }
        if (o.takes_value) left += " arg";
        std::string text = o.help;
        size_t pos = 0; bool first_line = true;
        while (true) {
            const size_t nl = text.find('\n', pos);
            const std::string line = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
            if (first_line) { std::cerr << left; for (size_t k = left.size(); k < 26; ++k) std::cerr << ' '; if (left.size() >= 26) std::cerr << "\n" << std::string(26, ' '); }
            else std::cerr << std::string(26, ' ');
            std::cerr << line << "\n";
            first_line = false;
            if (nl == std::string::npos) break;
            pos = nl + 1;
        }
    }
    std::cerr << "\n";
}

struct Parsed { std::map<std::string, std::string> values; };   // long name -> value ("" for switches)

const OptSpec* find_long(const std::string& name)
{
    const OptSpec* exact = nullptr; const OptSpec* guess = nullptr; int n_guess = 0;
    for (const OptSpec& o : kSpecs) {
        if (name == o.longname) exact = &o;
        else if (std::strncmp(o.longname, name.c_str(), name.size()) == 0) { guess = &o; ++n_guess; }
    }
    if (exact) return exact;
    if (n_guess == 1) return guess;
    if (n_guess > 1) throw std::runtime_error("option '--" + name + "' is ambiguous");
    throw std::runtime_error("unrecognised option '--" + name + "'");
}

const OptSpec* find_short(char c)
{
    for (const OptSpec& o : kSpecs) if (o.shortname == c) return &o;
    throw std::runtime_error(std::string("unrecognised option '-") + c + "'");
}

void store(Parsed& p, const OptSpec* o, const std::string& v)
{
    const std::string key = o->longname;
    if (p.values.count(key)) throw std::runtime_error("option '--" + key + "' cannot be specified more than once");
    p.values[key] = v;
}

Parsed parse_command_line(int argc, char** argv)
{
    Parsed p;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            const OptSpec* o = find_long(name);
            if (!o->takes_value) {
                if (eq != std::string::npos) throw std::runtime_error(std::string("option '--") + o->longname + "' does not take any arguments");
                store(p, o, "");
            } else if (eq != std::string::npos) store(p, o, a.substr(eq + 1));
            else {
                if (i + 1 >= argc) throw std::runtime_error(std::string("the required argument for option '--") + o->longname + "' is missing");
                store(p, o, argv[++i]);
            }
        } else if (a.size() >= 2 && a[0] == '-' && a != "--") {
            for (size_t k = 1; k < a.size(); ++k) {          // -v, -i value, -ivalue, grouped switches
                const OptSpec* o = find_short(a[k]);
                if (!o->takes_value) { store(p, o, ""); continue; }
                if (k + 1 < a.size()) store(p, o, a.substr(k + 1));
                else {
                    if (i + 1 >= argc) throw std::runtime_error(std::string("the required argument for option '--") + o->longname + "' is missing");
                    store(p, o, argv[++i]);
                }
                break;
            }
        } else {
            throw std::runtime_error("too many positional options have been specified on the command line");
        }
    }
    return p;
}

template <class T>
T to_integer(const std::string& v, const char* optname)
{
    try {
        size_t used = 0;
        const long long x = std::stoll(v, &used);
        if (used != v.size() || (std::is_unsigned<T>::value && x < 0)) throw std::invalid_argument(v);
        return static_cast<T>(x);
    } catch (const std::exception&) {
        throw std::runtime_error("the argument ('" + v + "') for option '--" + optname + "' is invalid");
    }
}

enum Modes { BASE = 0, FASTA = 1, PAIRED = 2, HASH = 4 };      // main.cpp:15-21

struct Options {                                               // main.cpp:28-38
    int mode = BASE;
    ssize_t memLimit = 2L * 1024L * 1024L * 1024L;
    std::string input_1, input_2, output_1, output_2;
    unsigned hammdist = 2;
    bool unordered = false, verbose = false, write_clusters = false;
};

bool parse_args(int argc, char** argv, Options& opts)          // main.cpp:40-179
{
    try {
        const Parsed vm = parse_command_line(argc, argv);
        auto count = [&](const char* k) { return vm.values.count(k) ? 1 : 0; };
        if (count("help")) { print_help(); return false; }     // main.cpp:85-90: to stderr, exit code 1
        for (const char* req : {"input-1", "output-1"})          // po::notify: required options
            if (!count(req)) throw std::runtime_error(std::string("the option '--") + req + "' is required but missing");
        opts.verbose = count("verbose"); opts.write_clusters = count("write-clusters"); opts.unordered = count("unordered");
        const bool hash_opt = count("fast");
        opts.input_1 = vm.values.at("input-1"); opts.output_1 = vm.values.at("output-1");
        if (count("input-2")) opts.input_2 = vm.values.at("input-2");
        if (count("output-2")) opts.output_2 = vm.values.at("output-2");
        if (count("distance")) opts.hammdist = to_integer<unsigned>(vm.values.at("distance"), "distance");
        ssize_t mem_value = 0;
        if (count("mem-limit")) mem_value = to_integer<ssize_t>(vm.values.at("mem-limit"), "mem-limit");

        if (count("input-2") ^ count("output-2"))                                                    // :94-95
            throw std::runtime_error("Both input-2 and output-2 arguments are required for paired-end mode!");
        if (count("input-2")) opts.mode |= PAIRED;                                                  // :98-99
        if (count("input-2")) {                                                                      // :102-108
            if (opts.input_1 == opts.input_2) throw std::runtime_error("Paired input files should not be the same file!");
            if (opts.output_1 == opts.output_2) throw std::runtime_error("Paired output files should not be the same file!");
        }
        if (count("format")) {                                                                       // :111-120
            const std::string& v = vm.values.at("format");
            if (v == "fastq") {}
            else if (v == "fasta") opts.mode |= FASTA;
            else throw std::runtime_error("Only \"fastq\" or \"fasta\" file formats are supported!");
        }
        if (count("compare-seq")) {                                                                  // :123-134
            const std::string& v = vm.values.at("compare-seq");
            if (v != "tight" && v != "loose" && v != "tail-hamming") throw std::runtime_error("Unsupported compare-seq type provided!");
        }
        if (count("mem-limit")) {                                                                    // :137-144
            if (mem_value >= 500L && mem_value <= 10240L) opts.memLimit = mem_value * 1024L * 1024L;
            else throw std::runtime_error("Value of unsupported range provided for --mem-limit option!");
        }
        if (hash_opt) {                                                                              // :147-155
            opts.mode |= HASH;
            if (count("compare-seq") || count("distance") || opts.write_clusters)
                throw std::runtime_error("--fast mode was enabled, but argument(s) for sequence-based mode were provided!");
        }
        if (opts.unordered) {                                                                        // :158-164
            if (!hash_opt) throw std::runtime_error("--unordered argument can only be used with --fast mode!");
            if (!count("input-2")) throw std::runtime_error("--unordered argument can only be used with paired inputs!");
        }
    } catch (const std::exception& e) {                                                              // :167-172
        std::cerr << "An error occured during arguments parsing:\n" << e.what() << '\n';
        return false;
    }
    return true;
}

} // namespace

int main(int argc, char** argv)                                // main.cpp:181-262
{
    // FQD_HOST_TIMING: where the wall time of the process goes that no stage of the run accounts for — before main
    // (the loader, the HIP runtime's own start) is what the caller's clock saw and these two lines did not
    const auto t_main = std::chrono::steady_clock::now();
    struct AtEnd {
        std::chrono::steady_clock::time_point t0;
        ~AtEnd() {
            if (std::getenv("FQD_HOST_TIMING"))
                std::cerr << "[host timing] process: main() took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
                          << " s (what follows it: the runtime and the driver giving the device memory back)\n";
        }
    } at_end{t_main};
    Options opts;
    if (!parse_args(argc, argv, opts)) return 1;
    try {
        if (!(opts.mode & HASH))
            throw std::runtime_error("this build implements the hash-based --fast mode only (MI355X engine); "
                                     "run the sequence-based modes with the reference fastq-dupaway");
        fqdhost::Tuning tune;
        if (const char* d = std::getenv("FQD_DEVICE")) tune.device = std::atoi(d);
        tune.devices = fqdhost::devices_from_env();
        if (const char* x = std::getenv("FQD_EXCHANGE")) tune.use_rccl = std::string(x) != "copy";
        if (const char* j = std::getenv("FQD_FULL_JOIN")) tune.reference_tail_rule = !(j[0] == '1');
        if (const char* b = std::getenv("FQD_BLOCK_MB")) { const long mb = std::atol(b); if (mb > 0) tune.block_bytes = static_cast<size_t>(mb) << 20; }
        tune.leave_memory_to_exit = std::getenv("FQD_FREE_AT_END") == nullptr;   // the process ends after this run
        fqdhost::TemporaryDirectory tempdir;                   // main.cpp:192 (created lazily here)
        const fqdhost::Format fmt = (opts.mode & FASTA) ? fqdhost::Format::Fasta : fqdhost::Format::Fastq;
        fqdhost::HashDupRemover remover(fmt, opts.memLimit, &tempdir, opts.verbose, tune);   // main.cpp:218-242
        if (opts.mode & PAIRED) remover.filterPE(opts.input_1, opts.input_2, opts.output_1, opts.output_2, opts.unordered);
        else                    remover.filterSE(opts.input_1, opts.output_1);
    } catch (const std::exception& exc) {                      // main.cpp:250-254
        std::cerr << "An error occured during fastq-dupaway execution:\n" << exc.what() << '\n';
        return 1;
    } catch (...) {
        std::cerr << "Unknown error occured during fastq-dupaway execution!\n";
        return 1;
    }
    return 0;
}
