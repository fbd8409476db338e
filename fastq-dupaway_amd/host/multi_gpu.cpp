#include "multi_gpu.hpp"

#include <cstdlib>
#include <stdexcept>
#include <string>

namespace fqdhost {

std::vector<int> devices_from_env()
{
    std::vector<int> out;
    const char* v = std::getenv("FQD_DEVICES");
    if (!v || !*v) return out;
    std::string s(v);
    size_t at = 0;
    while (at <= s.size()) {
        const size_t comma = s.find(',', at);
        const std::string item = s.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
        if (item.empty() || item.find_first_not_of("0123456789") != std::string::npos)
            throw std::runtime_error("FQD_DEVICES: a comma-separated list of GPU ordinals is expected, got '" + s + "'");
        out.push_back(std::atoi(item.c_str()));
        if (comma == std::string::npos) break;
        at = comma + 1;
    }
    return out;
}

} // namespace fqdhost
