// lambda_ext.hpp -- thin C++ wrapper over the C ABI: RAII handle, errors rethrown as std::runtime_error (what the
// reference's callers expect: failures in the search path surface as exceptions, /root/reference/src/search.cpp:98-125).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/lambda_ext.h"

namespace lambda_amd
{

class Engine
{
public:
    explicit Engine(int device = 0)
    {
        if (int rc = lx_create(device, &h_); rc != LX_OK)
            throw std::runtime_error(std::string("lambda_ext: ") + lx_last_error(nullptr));
    }
    ~Engine() { lx_destroy(h_); }
    Engine(Engine const &)             = delete;
    Engine & operator=(Engine const &) = delete;

    lx_handle * raw() { return h_; }

    void check(int rc) const
    {
        if (rc != LX_OK)
            throw std::runtime_error(std::string("lambda_ext: ") + lx_last_error(h_));
    }
    void setScoring(lx_scoring const & sc, int slot = 0) { check(lx_set_scoring(h_, slot, &sc)); }
    void setOption(int opt, uint64_t v) { check(lx_set_option(h_, opt, v)); }

    // _performAlignment<false>, src/search_algo.hpp:1246
    void score(int slot, uint8_t const * q, uint64_t qBytes, uint8_t const * s, uint64_t sBytes,
               std::vector<lx_extension> const & ext, std::vector<int32_t> & out)
    {
        out.assign(ext.size(), 0);
        check(lx_score_batch(h_, slot, q, qBytes, s, sBytes, ext.data(), ext.size(), out.data()));
    }
    // _performAlignment<true>, src/search_algo.hpp:1296
    void align(int slot, uint8_t const * q, uint64_t qBytes, uint8_t const * s, uint64_t sBytes,
               std::vector<lx_extension> const & ext, std::vector<lx_hsp> & hsp, std::vector<uint8_t> & ops,
               std::vector<uint64_t> & opsOff)
    {
        opsOff.resize(ext.size());
        uint64_t total = 0;
        for (size_t i = 0; i < ext.size(); ++i)
        {
            opsOff[i] = total;
            total += (uint64_t)ext[i].q_len + ext[i].s_len;
        }
        ops.assign(total + 1, 0);
        hsp.assign(ext.size(), lx_hsp{});
        check(lx_align_batch(h_, slot, q, qBytes, s, sBytes, ext.data(), ext.size(), nullptr, hsp.data(), ops.data(), opsOff.data()));
    }

private:
    lx_handle * h_ = nullptr;
};

} // namespace lambda_amd
