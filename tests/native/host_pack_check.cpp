// host_pack_check — runs the SAME packer the kernels use (fqd_device.hpp; its two
// byte-shuffle instructions have host bodies for exactly this purpose) on the host
// and prints, per input line, "<nwords> <hash> <bad_pos> <bad_byte> <word>...".
// tests/test_packer_host.py compares the words with an independent numpy packing
// and checks injectivity against the oracle's base-5 keys.  No GPU involved.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>
#include "../../fastq-dupaway_amd/csrc/fqd_device.hpp"

int main()
{
    std::string line;
    while (std::getline(std::cin, line)) {
        // input line: <shift 0..3> <hex-escaped bytes>
        unsigned shift = line[0] - '0';
        std::string raw;
        for (size_t i = 2; i + 1 < line.size(); i += 2) raw.push_back(char(std::stoi(line.substr(i, 2), nullptr, 16)));
        // lay the sequence `shift` bytes into an aligned dword buffer, garbage around it
        std::vector<uint32_t> buf((raw.size() + shift + 3) / 4 + 2, 0x5A5A5A5Au);
        std::memcpy(reinterpret_cast<char*>(buf.data()) + shift, raw.data(), raw.size());
        const uint32_t len = uint32_t(raw.size());
        std::vector<uint64_t> words;
        uint64_t h = fqd::hash_begin(len, 0);
        auto sink = [&](uint64_t w) { words.push_back(w); h = fqd::hash_word(h, w); };
        const uint32_t diff = fqd::pack_mate(buf.data(), shift, len, sink);
        uint32_t bad_byte = 0, bad_pos = 0xFFFFFFFFu;
        if (diff) bad_pos = fqd::first_bad_base(reinterpret_cast<const uint8_t*>(raw.data()), len, &bad_byte);
        h = fqd::hash_end(h);
        std::printf("%zu %llu %u %u", words.size(), (unsigned long long)h, bad_pos, bad_byte);
        for (uint64_t w : words) std::printf(" %llu", (unsigned long long)w);
        std::printf("\n");
    }
    return 0;
}
