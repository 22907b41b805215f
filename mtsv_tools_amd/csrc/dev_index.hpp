// dev_index.hpp -- owning handle of the HBM-resident index on one device.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "dev_layout.hpp"
#include "mgindex.hpp"

namespace mtsv {

[[noreturn]] void throw_hip(hipError_t e, const char* what, const char* file, int line);
#define HIP_CHECK(expr)                                              \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) ::mtsv::throw_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

struct DeviceIndex {
    int device = -1;
    RankBlock* d_blocks = nullptr;
    uint32_t* d_sa_sample = nullptr;
    uint32_t* d_sa_full = nullptr;
    uint8_t* d_text = nullptr;
    uint32_t* d_bin_end = nullptr;
    DevBin* d_bins = nullptr;
    uint32_t* d_bin_lut = nullptr;
    uint2* d_kmer = nullptr;
    DevIndexView view{};
    uint64_t bytes = 0;
    uint32_t flags = 0;
    float accel_build_ms = 0.f;

    DeviceIndex() = default;
    DeviceIndex(const DeviceIndex&) = delete;
    DeviceIndex& operator=(const DeviceIndex&) = delete;
    ~DeviceIndex();
};

// Pack (host, multi-threaded) + upload + build the HBM-only acceleration structures.
// Throws std::runtime_error ("device: ...", "limit: ...", "format: ...").
std::unique_ptr<DeviceIndex> upload_index(const HostIndex& hx, int device, uint32_t flags);

}  // namespace mtsv

// the opaque C handle
struct mtsv_cached_workspace {  // workspace kept between mtsv_bin_batch* calls; mu held while a call runs on it
    std::mutex mu;
    struct mtsv_batch* b = nullptr;
};
struct mtsv_index {
    mtsv::HostIndex host;
    std::mutex mu;  // guards the two maps (never held while a batch runs)
    std::map<int, std::unique_ptr<mtsv::DeviceIndex>> dev;
    // (device, slot): slot > 0 when one call lists a device more than once (mtsv_bin_batch_multi)
    std::map<std::pair<int, int>, std::unique_ptr<mtsv_cached_workspace>> cached_batch;
};
