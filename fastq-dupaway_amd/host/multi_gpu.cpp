#include "multi_gpu.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <stdexcept>
#include <string>

#include <rccl/rccl.h>
#include <unistd.h>

namespace fqdhost {

void ExchangePlan::finish()
{
    send_off.assign(size_t(n) * n, 0); recv_off.assign(size_t(n) * n, 0);
    n_send.assign(n, 0); n_recv.assign(n, 0);
    for (int s = 0; s < n; ++s) {
        uint64_t at = 0;
        for (int d = 0; d < n; ++d) { send_off[size_t(s) * n + d] = at; at += send[size_t(s) * n + d]; }
        n_send[s] = at;
    }
    for (int d = 0; d < n; ++d) {
        uint64_t at = 0;
        for (int s = 0; s < n; ++s) { recv_off[size_t(d) * n + s] = at; at += send[size_t(s) * n + d]; }
        n_recv[d] = at;
    }
}

std::vector<Transfer> forward_transfers(const ExchangePlan& p, const std::vector<const void*>& grouped,
                                        const std::vector<void*>& received, size_t item_bytes)
{
    std::vector<Transfer> t;
    for (int s = 0; s < p.n; ++s)
        for (int d = 0; d < p.n; ++d) {
            const uint64_t c = p.send[size_t(s) * p.n + d];
            if (!c) continue;
            t.push_back({s, static_cast<const char*>(grouped[s]) + p.send_off[size_t(s) * p.n + d] * item_bytes,
                         d, static_cast<char*>(received[d]) + p.recv_off[size_t(d) * p.n + s] * item_bytes, c * item_bytes});
        }
    return t;
}

std::vector<Transfer> backward_transfers(const ExchangePlan& p, const std::vector<const void*>& at_owner,
                                         const std::vector<void*>& at_source, size_t item_bytes)
{
    std::vector<Transfer> t;
    for (int d = 0; d < p.n; ++d)
        for (int s = 0; s < p.n; ++s) {
            const uint64_t c = p.send[size_t(s) * p.n + d];
            if (!c) continue;
            t.push_back({d, static_cast<const char*>(at_owner[d]) + p.recv_off[size_t(d) * p.n + s] * item_bytes,
                         s, static_cast<char*>(at_source[s]) + p.send_off[size_t(s) * p.n + d] * item_bytes, c * item_bytes});
        }
    return t;
}

namespace {

void hip_ok(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Peer copies: every transfer is one asynchronous copy on the RECEIVING rank's stream.
class CopyExchange : public Exchange {
public:
    CopyExchange(const std::vector<int>& devices, const std::vector<hipStream_t>& streams) : dev_(devices), st_(streams)
    {
        for (size_t a = 0; a < dev_.size(); ++a)
            for (size_t b = 0; b < dev_.size(); ++b)
                if (dev_[a] != dev_[b]) {
                    int can = 0;
                    hip_ok(hipDeviceCanAccessPeer(&can, dev_[a], dev_[b]), "hipDeviceCanAccessPeer");
                    if (can) { hip_ok(hipSetDevice(dev_[a]), "hipSetDevice"); (void)hipDeviceEnablePeerAccess(dev_[b], 0); }   // already enabled is fine
                }
    }
    void run(const std::vector<Transfer>& t) override
    {
        for (const Transfer& x : t) {
            hip_ok(hipSetDevice(dev_[x.dst_rank]), "hipSetDevice");
            if (dev_[x.src_rank] == dev_[x.dst_rank])
                hip_ok(hipMemcpyAsync(x.dst, x.src, x.bytes, hipMemcpyDeviceToDevice, st_[x.dst_rank]), "hipMemcpyAsync");
            else
                hip_ok(hipMemcpyPeerAsync(x.dst, dev_[x.dst_rank], x.src, dev_[x.src_rank], x.bytes, st_[x.dst_rank]), "hipMemcpyPeerAsync");
        }
    }
    const char* name() const override { return "peer copies"; }
private:
    std::vector<int> dev_; std::vector<hipStream_t> st_;
};

// RCCL: one communicator per rank in this process (ncclCommInitAll); an exchange is ONE group of
// sends and receives, i.e. a direct all-to-all over the xGMI links, not a ring.
class RcclExchange : public Exchange {
public:
    RcclExchange(const std::vector<int>& devices, const std::vector<hipStream_t>& streams) : dev_(devices), st_(streams), comm_(devices.size())
    {
        // RCCL announces itself on stdout when NCCL_DEBUG asks for it; stdout belongs to the -v summary
        // lines (hash_dup_remover.hpp:146-147), so the announcement is sent to stderr
        std::fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        const ncclResult_t r = ncclCommInitAll(comm_.data(), static_cast<int>(dev_.size()), dev_.data());
        std::fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); (void)close(saved); }
        if (r != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    }
    ~RcclExchange() override { for (ncclComm_t c : comm_) if (c) (void)ncclCommDestroy(c); }
    void run(const std::vector<Transfer>& t) override
    {
        // A message above 1 GiB does not arrive whole on this image (RCCL 2.26.6: only its first half,
        // tools/a2a_probe.py): messages are cut into pieces of at most kPiece bytes, each piece its
        // own matched send/receive pair inside the group.
        constexpr size_t kPiece = size_t(512) << 20;
        check(ncclGroupStart(), "ncclGroupStart");
        for (const Transfer& x : t)
            for (size_t at = 0; at < x.bytes; at += kPiece) {
                const size_t m = std::min(kPiece, x.bytes - at);
                check(ncclSend(static_cast<const char*>(x.src) + at, m, ncclUint8, x.dst_rank, comm_[x.src_rank], st_[x.src_rank]), "ncclSend");
                check(ncclRecv(static_cast<char*>(x.dst) + at, m, ncclUint8, x.src_rank, comm_[x.dst_rank], st_[x.dst_rank]), "ncclRecv");
            }
        check(ncclGroupEnd(), "ncclGroupEnd");
    }
    const char* name() const override { return "RCCL all-to-all"; }
private:
    static void check(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
    }
    std::vector<int> dev_; std::vector<hipStream_t> st_; std::vector<ncclComm_t> comm_;
};

} // namespace

std::unique_ptr<Exchange> Exchange::create(const std::vector<int>& devices, const std::vector<hipStream_t>& streams, bool prefer_rccl)
{
    const bool distinct = std::set<int>(devices.begin(), devices.end()).size() == devices.size();
    if (prefer_rccl && distinct) return std::make_unique<RcclExchange>(devices, streams);
    return std::make_unique<CopyExchange>(devices, streams);
}

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
